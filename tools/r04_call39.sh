#!/bin/bash
# MI355_PF_QKV_IN_FOLD (mask 256): the fold's spare blocks touch the next layer's QKV weights.  Equality test, step A/B, per-kernel times.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fold_touch" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 )
for b in 64 16 8; do for pf in 0 256 0 256; do
  timeout 300 python bench.py --batch $b --no-sweep --no-cpu-baseline --steps 64 --warmup 8 --prefetch $pf 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b=$b prefetch=$pf', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done 2>&1 | tee gpurun_out/r04/fold_touch_ab.txt
export ROUND=r04
bash tools/trace_bench.sh pf256_b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --prefetch 256
grep -E "fullk64|add_rmsnorm" gpurun_out/r04/kernel_stats_pf256_b64.txt | cut -c1-150 | tee -a gpurun_out/r04/fold_touch_ab.txt
