#!/bin/bash
mkdir -p gpurun_out/r03
for v in 0 1 2 3; do
  echo "=== var $v"; python tools/gemm_bench.py --ms 64 --shapes gate_up --var $v 2>&1 | tail -3
done > gpurun_out/r03/wide_dbg_vars.txt 2>&1
cat gpurun_out/r03/wide_dbg_vars.txt
