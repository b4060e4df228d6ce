#!/bin/bash
# MI355_PF_DOWN_IN_O (mask 512): the O launch's spare CUs read the down weights.  256 = the default (QKV weights from the fold), 768 = both.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fold_touch" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 )
for b in 64 16 8; do for pf in 256 768 256 768; do
  timeout 300 python bench.py --batch $b --no-sweep --no-cpu-baseline --steps 64 --warmup 8 --prefetch $pf 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b=$b prefetch=$pf', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done 2>&1 | tee gpurun_out/r04/down_in_o_ab.txt
for cap in 2 4 16; do
  timeout 300 python bench.py --batch 64 --no-sweep --no-cpu-baseline --steps 64 --warmup 8 --prefetch 768 --debug-set 2=$cap 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b=64 prefetch=768 lines per thread=$cap', d['ms_per_step'], d['ms_per_step_repeats'])"
done 2>&1 | tee -a gpurun_out/r04/down_in_o_ab.txt
export ROUND=r04
bash tools/trace_bench.sh pf768_b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --prefetch 768
bash tools/trace_bench.sh pf768_b8 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --prefetch 768 --batch 8
for t in pf768_b64 pf768_b8; do head -12 gpurun_out/r04/kernel_stats_$t.txt | grep "fullk64\|splitk64\|wide" | cut -c1-150; done | tee -a gpurun_out/r04/down_in_o_ab.txt
