#!/bin/bash
mkdir -p gpurun_out/r03
python tools/prefill_gemm_bench.py --ms 128,256,512,1024,2048,4096 2>&1 | grep -v amdgpu > gpurun_out/r03/prefill_gemm_final.txt; cat gpurun_out/r03/prefill_gemm_final.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -x -q -k "prefill or large_batch" 2>&1 | tail -2
