#!/bin/bash
mkdir -p gpurun_out/r03
{
echo "== fp16 reference (engine path: partial for qkv/o/down)"; python tools/gemm_bench.py --ms 64 --partial 1 2>&1 | grep -v amdgpu
for nbw in 0 10 11 13 7; do
  echo "== bf16 cfg override nbw=$nbw (0 = planner: cfg 8)"; python tools/gemm_bench.py --ms 64 --partial 1 --bf16 1 --nbw $nbw 2>&1 | grep -v amdgpu
done
} > gpurun_out/r03/bf16_gemm_m64_shapes.txt 2>&1
cat gpurun_out/r03/bf16_gemm_m64_shapes.txt
