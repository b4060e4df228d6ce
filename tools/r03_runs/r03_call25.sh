#!/bin/bash
# A/B on one box: tuning library built from the previous gemm.hip (ds_bpermute meta) vs product library (meta loaded in the accumulator layout)
for rep in 1 2; do
echo "== OLD (tuning lib, previous source) rep $rep"; python tools/gemm_bench.py --ms 1,16,32 --partial 1 --iters 96 2>&1 | grep -v amdgpu
echo "== NEW (product lib) rep $rep"; python tools/gemm_bench.py --product --ms 1,16,32 --partial 1 --iters 96 2>&1 | grep -v amdgpu
done
echo "== bf16 OLD"; python tools/gemm_bench.py --ms 16,64 --partial 1 --bf16 1 --iters 96 2>&1 | grep -v amdgpu
echo "== bf16 NEW"; python tools/gemm_bench.py --product --ms 16,64 --partial 1 --bf16 1 --iters 96 2>&1 | grep -v amdgpu
