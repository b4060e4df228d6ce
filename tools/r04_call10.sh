#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fused_linear.py -x -q 2>&1 | tail -4 | tee gpurun_out/r04_c10_tests.txt
ROUND=r04 bash tools/trace_bench.sh b64 --steps 20 --warmup 5 --no-sweep --no-cpu-baseline
head -12 gpurun_out/r04/kernel_stats_b64.txt
