#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fused_linear.py -x -q -k "partial_img or wide_gemm_writes" 2>&1 | tail -8 | tee gpurun_out/r04_c8_tests.txt
timeout 300 python tools/splitk64_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c8_splitk64.txt
timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py -x -q -k "engine or step or prefill or generate" 2>&1 | tail -5 | tee gpurun_out/r04_c8_tests_engine.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b64', d['ms_per_step'], [ (s['batch'], s['ms_per_step']) for s in d['sweep']], {k:v.get('ms_per_step') for k,v in d['other_workloads'].items()}, d['step_roofline']['eager_kernel_ms_per_step'])" | tee gpurun_out/r04_c8_bench.txt
