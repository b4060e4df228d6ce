#!/bin/bash
# round-3 evidence, part A: per-kernel traces (b=64, b=1) under rocprofv3 --kernel-trace --stats
export ROUND=r03
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
bash $R/tools/trace_bench.sh b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
bash $R/tools/trace_bench.sh b1 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 1
head -14 $O/kernel_stats_b64.txt
