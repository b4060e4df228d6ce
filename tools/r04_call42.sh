#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python tools/wide_img_time.py --ms 8,16,32,64 --resident 2>&1 | grep -v amdgpu.ids | grep -v "row-major" | tee gpurun_out/r04/gate_up_resident.txt
