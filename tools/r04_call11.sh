#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in 0 1; do timeout 120 python tools/attn_bench.py --product --copies $c 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r04_c11_attn_mall.txt
timeout 120 python tools/attn_bench.py --product --copies 1 --batch 32 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_c11_attn_mall.txt
timeout 120 python tools/attn_bench.py --product --copies 0 --batch 32 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_c11_attn_mall.txt
timeout 120 python tools/attn_bench.py --product --copies 1 --int8 --ctx 4096 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_c11_attn_mall.txt
timeout 120 python tools/attn_bench.py --product --copies 0 --int8 --ctx 4096 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_c11_attn_mall.txt
