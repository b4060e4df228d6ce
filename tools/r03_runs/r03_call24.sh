#!/bin/bash
echo "== fp16 M=1,8,16,32 (partial: engine path)"; python tools/gemm_bench.py --ms 1,16,32 --partial 1 2>&1 | grep -v amdgpu
echo "== bf16 M=16,64"; python tools/gemm_bench.py --ms 16,64 --partial 1 --bf16 1 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_baseline_shapes.py -x -q -k "linear" 2>&1 | tail -2
