#!/bin/bash
# one rank's TP step with the final sources: qwen2-7b tp 2 / 4 (28 heads: no tp 8), llama3-70b tp 8, qwen2-72b tp 8 (configs[4] target: b = 8 decode and the 40-row verify shape), + kernel traces
cd $GRAFT_REPO_ROOT; export ROUND=r05 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'tok/s per rank-step', 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
( for so in 2 4; do python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b   one rank of tp$so b=64"; done
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b one rank of tp8 b=32"
  python bench.py --workload qwen2-72b-w4a16 --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "qwen2-72b  one rank of tp8 b=8 "
  python bench.py --workload qwen2-72b-w4a16 --shard-of 8 --batch 40 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "qwen2-72b  one rank of tp8 b=40"
) 2>&1 | tee $O/tp_shard_final.txt
bash tools/trace_bench.sh tp4_shard_b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --shard-of 4
