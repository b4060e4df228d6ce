#!/bin/bash
cd $GRAFT_REPO_ROOT; export ROUND=r04; mkdir -p gpurun_out/r04
bash tools/trace_bench.sh b8 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 8
bash tools/trace_bench.sh b1 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 1
bash tools/trace_bench.sh b16 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 16
head -9 gpurun_out/r04/kernel_stats_b8.txt; head -10 gpurun_out/r04/kernel_stats_b1.txt; head -9 gpurun_out/r04/kernel_stats_b16.txt
