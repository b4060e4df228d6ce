#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in 4 6 8 12; do echo "fuse_rows=$t"; timeout 300 python tools/batch_sweep.py --batches 5,6,7,8,9,10,12 --tune 6=$t 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r04_c14_sweep_small.txt
