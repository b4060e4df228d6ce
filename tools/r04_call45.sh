#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python tools/fullk_time.py --product --resident --ms 1,4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/fullk_resident.txt
