#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
for b in 64 48 32; do for m in 0 128 256 384 0 384; do
  timeout 300 python bench.py --no-cpu-baseline --no-sweep --batch $b --prefetch $m --steps 64 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch', d['config']['batch'], 'ride mask', $m, 'ms/step', d['ms_per_step'], 'p50', d['p50_ms'])"
done; done 2>&1 | tee $O/ride_ab.txt
