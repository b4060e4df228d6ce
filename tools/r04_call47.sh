#!/bin/bash
# wide GEMM: K-slice merge in one round at every row-block count (four row blocks: a wave keeps the row block of its K slice in registers)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
( timeout 1200 python -m pytest tests/test_gpu_fused_linear.py tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -m gpu -k "wide or deferred or image or img or silu or engine or bf16 or linear" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 )
timeout 600 python tools/wide_img_time.py --ms 64,48,32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/wide_merge_one_round.txt
timeout 600 python tools/batch_sweep.py --batches 32,40,48,64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/wide_merge_one_round.txt
