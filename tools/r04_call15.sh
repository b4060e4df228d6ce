#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in 99 2 4; do echo "fuse_rows=$t"; timeout 300 python tools/batch_sweep.py --batches 1,2,3,4,5 --tune 6=$t 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r04_c15_sweep_small.txt
