#!/bin/bash
# (1) bound of a down_proj weight prefetch: one cache-resident weight copy; same for gate_up.  (2) delay of the fold's touch blocks.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python tools/splitk64_time.py --ms 64,16 --resident 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee gpurun_out/r04/down_resident.txt
for d in 0 2 4 8; do
  timeout 300 python bench.py --batch 64 --no-sweep --no-cpu-baseline --steps 64 --warmup 8 --prefetch 256 --debug-set 3=$d 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b=64 prefetch=256 delay=$d', d['ms_per_step'], d['ms_per_step_repeats'])"
done 2>&1 | tee gpurun_out/r04/fold_touch_delay.txt
