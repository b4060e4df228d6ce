#!/bin/bash
# an instance per row-block count (1 / 2 / 3 / 4 MFMAs per unit) in gemm_fullk64 / gemm_wide / gemm_splitk64: parity, gate_up micro A/B, sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
( timeout 1200 python -m pytest tests/test_gpu_fused_linear.py tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 ) | tee gpurun_out/r04/mblk_tests.txt
timeout 600 python tools/wide_img_time.py --ms 8,16,40,48 --tuning --dbg 0,17 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/wide_mblk.txt
timeout 600 python tools/batch_sweep.py --batches 5,8,12,16,17,24,32,33,40,48,49,56,64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/batch_sweep_mblk.txt
