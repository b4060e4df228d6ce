#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/wide_img_time.py --ms 64,48,32,17 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c6_wide_img.txt
timeout 300 python tools/wide_img_time.py --ms 64,32 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_c6_wide_img.txt
