#!/bin/bash
# fold touch: one dword per 128 bytes (default) vs per 64 bytes (--debug-set 3=100, tuning build) -- does a dword bring the whole L2 line into the caches?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for rep in 1 2 3; do for v in 0 100; do
  timeout 300 python bench.py --batch 64 --no-sweep --no-cpu-baseline --steps 64 --warmup 8 --debug-set 3=$v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b=64 tuning build, touch step', 64 if $v==100 else 128, 'B:', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done 2>&1 | tee gpurun_out/r04/fold_touch_step.txt
