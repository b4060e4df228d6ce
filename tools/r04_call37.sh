#!/bin/bash
# wide GEMM at <= 32 rows: K-slice merge in one round, no norm partial-sum requests from waves past the last row
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
( timeout 1200 python -m pytest tests/test_gpu_fused_linear.py tests/test_gpu_parity.py -x -q -m gpu -k "wide or deferred or image or img or silu or engine" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 )
timeout 600 python tools/wide_img_time.py --ms 8,16,32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/wide_merge1.txt
timeout 600 python tools/batch_sweep.py --batches 5,8,16,24,32,64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/batch_sweep_merge1.txt
