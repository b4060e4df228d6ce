#!/bin/bash
# Round 5, call B: the TP step with the QKV shard on the image launch -- multi-process engine tests on one GPU, then ONE rank's TP step
# (bench.py --shard-of: per-rank shapes + the fused all-reduce launches with a world-1 context) against the round-4 chain (tuning build, 5=2).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_allreduce.py -x -q -k "engine-2 or engine70-8 or kernels-2 or bf16-2" 2>&1 | tail -6
python -m pytest tests/test_gpu_tp_engine.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -4
( for so in 2 4; do
    python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('qwen2-7b shard-of $so image-qkv  ', d['ms_per_step'], d['roofline']['avg_launch_us'], d['step_roofline']['eager_kernel_ms_per_step'])"
    python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 --debug-set 5=2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('qwen2-7b shard-of $so round-4 chain', d['ms_per_step'], d['roofline']['avg_launch_us'], d['step_roofline']['eager_kernel_ms_per_step'])"
  done
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('llama3-70b shard-of 8 image-qkv  ', d['ms_per_step'], d['roofline']['avg_launch_us'], d['step_roofline']['eager_kernel_ms_per_step'])"
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 --debug-set 5=2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('llama3-70b shard-of 8 round-4 chain', d['ms_per_step'], d['roofline']['avg_launch_us'], d['step_roofline']['eager_kernel_ms_per_step'])"
) 2>&1 | tee $O/tp_shard_steps.txt
