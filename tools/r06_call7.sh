#!/bin/bash
cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step', 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
( python -m pytest tests/test_gpu_allreduce.py -m gpu -q -x -k "publish or engine7b or engine-2 or kernels-2" 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -6 )
( python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rope" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -4 )
for so in 2 4; do python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b   one rank of tp$so b=64"; done
python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b one rank of tp8 b=32"
bash tools/trace_bench.sh llama70b_tp8_shard_b32 --workload llama3-70b-awq --shard-of 8 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
head -9 $O/kernel_stats_llama70b_tp8_shard_b32.txt | cut -c1-150
