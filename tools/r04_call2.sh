#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/fullk64_time.py --ms 64 --variants "7=4;7=8;7=6;7=10;7=1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c2_time.txt
timeout 200 python tools/fullk_stamps.py --ms 64 --set 7=4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c2_stamps_dense.txt
