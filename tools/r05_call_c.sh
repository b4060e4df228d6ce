#!/bin/bash
# Round 5, call C: the all-reduce hand-over without the system-scope fences (write-through publishing stores) -- the multi-process checks in
# both forms, then ONE rank's TP step (bench.py --shard-of, world-1 context) new protocol vs MI355_AR_FULL_FENCES=1.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_allreduce.py -x -q -k "not engine70full" 2>&1 | tail -6
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'comm eager ms/step', d['step_roofline']['eager_kernel_ms_per_step']['comm'], 'gemm', d['step_roofline']['eager_kernel_ms_per_step']['gemm_quant'])"; }
( for so in 2 4; do
    python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b shard-of $so write-through"
    MI355_AR_FULL_FENCES=1 python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b shard-of $so full fences  "
  done
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b shard-of 8 write-through"
  MI355_AR_FULL_FENCES=1 python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b shard-of 8 full fences  "
) 2>&1 | tee $O/tp_allreduce_protocol.txt
