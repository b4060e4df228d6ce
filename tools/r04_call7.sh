#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 300 python tools/wide_img_time.py --ms 64,48,32,17 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c7_wide_img.txt
timeout 900 python -m pytest tests/test_gpu_fused_linear.py -x -q -k "deferred" 2>&1 | tail -4 | tee gpurun_out/r04_c7_tests_fused.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b64', d['ms_per_step'], [ (s['batch'], s['ms_per_step']) for s in d['sweep']], {k:v.get('ms_per_step') for k,v in d['other_workloads'].items()}, d['step_roofline']['eager_kernel_ms_per_step'])" | tee gpurun_out/r04_c7_bench.txt
