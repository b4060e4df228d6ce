#!/bin/bash
# upper bound of a weight prefetch for the full-K launches: one weight copy (cache-resident) against rotating HBM-resident copies
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python tools/fullk64_time.py --product --resident --ms 64,16 2>&1 | grep -v amdgpu.ids | grep -v "row-major\|composed" | tee gpurun_out/r04/fullk64_resident.txt
timeout 300 python tools/splitk64_time.py --help 2>&1 | tail -3
