#!/bin/bash
mkdir -p gpurun_out/r03
{
for ctx in 1024 4096; do
 for t in "" "6=3" "6=4"; do
  echo "=== int8 ctx $ctx tune [$t]"; python tools/attn_bench.py --int8 --ctx $ctx --tune "$t" 2>&1 | tail -1
 done
done
for t in "" "6=3" "6=4"; do echo "=== fp16 ctx 1024 tune [$t]"; python tools/attn_bench.py --ctx 1024 --tune "$t" 2>&1 | tail -1; done
} > gpurun_out/r03/attn_ng4.txt 2>&1
cat gpurun_out/r03/attn_ng4.txt
