#!/bin/bash
# Round 5, call J: the all-reduce with data-tagged granules (LL) for the <= 64-row calls: multi-process checks in all three forms, one rank's TP step LL vs write-through + flags.
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_allreduce.py -x -q -k "not engine70full" 2>&1 | tail -6
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'comm eager ms/step', d['step_roofline']['eager_kernel_ms_per_step']['comm'])"; }
( for so in 2 4; do for b in 64 8 1; do
    MI355_AR_LL=1 python bench.py --shard-of $so --batch $b --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b one rank of tp$so b=$b granules            "
    python bench.py --shard-of $so --batch $b --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b one rank of tp$so b=$b write-through + flags"
  done; done
  MI355_AR_LL=1 python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b one rank of tp8 b=32 granules            "
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b one rank of tp8 b=32 write-through + flags"
) 2>&1 | tee $O/tp_allreduce_granules.txt
