#!/bin/bash
# Round 5, call D: the TP step with the down_proj shard as K quarters (gate_up shard's SiLU output as an image): 7B-width two- / four-process
# engine tests, then ONE rank's TP step new vs without (tuning build, switch 5 = 4).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_allreduce.py -x -q -k "engine7b or engine-2 or engine70-8 or bf16-2" 2>&1 | tail -6
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
( for so in 2 4; do
    python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b shard-of $so down as K quarters"
    python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 --debug-set 5=4 2>/dev/null | tail -1 | line "qwen2-7b shard-of $so staged down     "
  done
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b shard-of 8 down as K quarters"
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 --debug-set 5=4 2>/dev/null | tail -1 | line "llama3-70b shard-of 8 staged down     "
) 2>&1 | tee $O/tp_down_quarters.txt
