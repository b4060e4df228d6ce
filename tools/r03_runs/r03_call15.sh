#!/bin/bash
# bf16 step at full width: per-kernel trace + timing at a few batch sizes (tools/bf16_step_bench.py)
mkdir -p gpurun_out/r03
python tools/bf16_step_bench.py > gpurun_out/r03/bf16_step.txt 2>&1
cat gpurun_out/r03/bf16_step.txt | grep -v amdgpu
