#!/bin/bash
# touch masks: 256 QKV weights from the fold's spare blocks, 512 O weights from the QKV launch's spare CUs, 1024 down weights from inside gate_up
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fold_touch" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 )
for b in 64 16; do for pf in 0 256 768 1280 1792 0 1792; do
  timeout 300 python bench.py --batch $b --no-sweep --no-cpu-baseline --steps 64 --warmup 8 --prefetch $pf 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b=$b prefetch=$pf', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done 2>&1 | tee gpurun_out/r04/touch_masks_ab.txt
export ROUND=r04
bash tools/trace_bench.sh pf1792_b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --prefetch 1792
head -14 gpurun_out/r04/kernel_stats_pf1792_b64.txt | grep GLOBAL | cut -c1-150 | tee -a gpurun_out/r04/touch_masks_ab.txt
