#!/bin/bash
mkdir -p gpurun_out/r03
for nbw in 0 5 9 12 10 11 7 13; do echo "== lm_head M=64 cfg override nbw=$nbw (0: planner)"; python tools/gemm_bench.py --ms 64,16,1 --shapes lm_head --nbw $nbw --iters 24 2>&1 | grep -v amdgpu; done > gpurun_out/r03/lmhead_shapes.txt 2>&1
cat gpurun_out/r03/lmhead_shapes.txt
