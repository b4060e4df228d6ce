#!/bin/bash
# end-of-round evidence for the sources as committed: full GPU suite, smoke(), then tools/collect_profiles_r04.sh (kernel traces, PMC traffic, sweeps, bench line)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -5 ) | tee gpurun_out/r04/full_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/smoke.txt
bash tools/collect_profiles_r04.sh > gpurun_out/r04/collect.log 2>&1
tail -1 gpurun_out/r04/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'], [(s['batch'], s['ms_per_step']) for s in d['sweep']])"
head -12 gpurun_out/r04/kernel_stats_b64.txt | cut -c1-160
