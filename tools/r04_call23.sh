#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee gpurun_out/r04_c23_bf16_tests.txt
timeout 600 python -m pytest tests/test_gpu_fused_linear.py tests/test_gpu_baseline_shapes.py -x -q -k "img or deferred or image or full_width" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b64', d['ms_per_step'], [ (s['batch'], s['ms_per_step']) for s in d['sweep']], {k:v.get('ms_per_step') for k,v in d['other_workloads'].items()})" | tee gpurun_out/r04_c23_bench.txt
