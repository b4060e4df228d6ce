#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for r in 1 2; do timeout 600 python tools/wide_img_time.py --ms 64 --tuning --dbg 0,18,0,18 2>&1 | grep -v amdgpu.ids | grep switch; done | tee gpurun_out/r04/wide_merge_ab.txt
