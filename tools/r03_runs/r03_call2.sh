#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
python tools/seam_bench.py --shapes qkv,o --nsplits 0,2,3,4,5,7 --cfgs 0,5,9,10,11,13 2>&1 | grep -v amdgpu.ids | tee $O/seam_qkv_o.txt
python tools/seam_bench.py --shapes down --nsplits 0,6,8,10,12,15 --cfgs 0,5,9,10,11 2>&1 | grep -v amdgpu.ids | tee $O/seam_down.txt
