#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_linear.py tests/test_gpu_baseline_shapes.py -x -q -k "residual or partial or per_rank or deferred or full_width" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 200 python tools/fullk64_time.py --product --ms 64,48 2>&1 | grep -v amdgpu.ids | grep default
timeout 200 python tools/fullk64_time.py --ms 64 --variants "4=3" 2>&1 | grep -v amdgpu.ids | grep -v composed
timeout 200 python tools/splitk64_time.py --ms 64,32 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b64', d['ms_per_step'], [ (s['batch'], s['ms_per_step']) for s in d['sweep']], {k:v.get('ms_per_step') for k,v in d['other_workloads'].items()})"
