#!/bin/bash
# A/B of the side-stream weight prefetch (tp = 1), b = 64 and b = 1: ms/step per mask
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
for b in 64 1 8 16; do
for m in 0 1 16 2 32 4 3 19 7 23 35 39 55; do
  python bench.py --no-cpu-baseline --no-sweep --batch $b --prefetch $m --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch', d['config']['batch'], 'prefetch mask', $m, 'ms/step', d['ms_per_step'], 'p50', d['p50_ms'])"
done; done 2>&1 | tee $O/prefetch_ab.txt
