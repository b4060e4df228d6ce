#!/bin/bash
mkdir -p gpurun_out/r03
{
for ctx in 1024 2048 4096; do
 for t in "" "6=5"; do
  echo "=== fp16 ctx $ctx tune [$t]"; timeout 120 python tools/attn_bench.py --ctx $ctx --tune "$t" 2>&1 | grep -v amdgpu | tail -3
 done
done
for b in 16 1; do for t in "" "6=5"; do echo "=== fp16 b $b ctx 1024 tune [$t]"; timeout 120 python tools/attn_bench.py --batch $b --tune "$t" 2>&1 | grep -v amdgpu | tail -3; done; done
} > gpurun_out/r03/attn_dma.txt 2>&1
cat gpurun_out/r03/attn_dma.txt
