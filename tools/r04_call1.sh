#!/bin/bash
# Round 4, GPU call 1: parity of the new 17-64-row full-K launches through the existing tests, then where their time goes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fused_linear.py -x -q 2>&1 | tail -8 | tee gpurun_out/r04_c1_tests.txt
timeout 300 python tools/fullk64_time.py --ms 64,32,17 --variants "5=2;6=1;6=2;7=1;7=2;7=3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c1_time.txt
timeout 200 python tools/fullk_stamps.py --ms 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c1_stamps_new.txt
timeout 200 python tools/fullk_stamps.py --ms 64 --set 5=2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c1_stamps_old.txt
timeout 200 python tools/fullk64_time.py --product --ms 64,48,32,17 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c1_time_product.txt
