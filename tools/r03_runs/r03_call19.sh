#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "3=2" "3=4"; do echo "== tiles per wave switch $t"; python tools/prefill_gemm_bench.py --ms 128,256,512,1024,2048,4096 --tune $t 2>&1 | grep -v amdgpu; done
} > gpurun_out/r03/prefill_gemm_tpw.txt 2>&1
cat gpurun_out/r03/prefill_gemm_tpw.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -x -q -k "prefill" 2>&1 | tail -2
