#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_call4.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_call4.txt; tail -25 $O/pytest_call4.txt
for b in 64 32 16; do for nm in "" "--no-meet"; do
  timeout 300 python bench.py --no-cpu-baseline --no-sweep --batch $b $nm --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch', d['config']['batch'], '$nm', 'ms/step', d['ms_per_step'], 'p50', d['p50_ms'], d.get('step_roofline',{}).get('eager_kernel_ms_per_step'))"
done; done 2>&1 | tee $O/meet_ab.txt
