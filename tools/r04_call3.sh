#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_linear.py -x -q 2>&1 | tail -8 | tee gpurun_out/r04_c3_tests_fused.txt
timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py -x -q -k "engine or step or prefill or generate" 2>&1 | tail -8 | tee gpurun_out/r04_c3_tests_engine.txt
timeout 300 python tools/fullk64_time.py --product --ms 64,48,32,17 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c3_time_product.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 | tee gpurun_out/r04_c3_bench.txt
