#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 1 2 4 8 16 32; do for ps in 0 256 512 1024; do timeout 100 python tools/attn_bench.py --batch $b --ps $ps 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r04_c30_attn_partition_sizes.txt
