#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python tools/wide_img_time.py --ms 8,16,32 --tuning --dbg 0,16,2,3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/wide_ring2.txt
