#!/bin/bash
mkdir -p gpurun_out/r03
{
for ctx in 2048 4096; do
 for t in "" "6=3"; do
  echo "=== int8 ctx $ctx tune [$t]"; python tools/attn_bench.py --int8 --ctx $ctx --tune "$t" 2>&1 | tail -2
 done
done
echo "=== fp16 ctx 2048"; python tools/attn_bench.py --ctx 2048 2>&1 | tail -2
} > gpurun_out/r03/attn_int8_ng.txt 2>&1
cat gpurun_out/r03/attn_int8_ng.txt
