#!/bin/bash
# full-K launches (helper's requests before the K waves', four-chunk ring for down): parity, kernels, then the step at small batches
timeout 600 python -m pytest tests/test_gpu_fused_linear.py -x -q 2>&1 | tail -3
timeout 400 python tools/fullk_time.py --sweep 5=0,2 --ms 1,4,8 2>&1 | grep -v amdgpu.ids
timeout 400 python tools/batch_sweep.py --batches 1,2,4,8,9,10,12,16 --tune 6=10 2>&1 | grep -v amdgpu.ids
timeout 400 python tools/batch_sweep.py --batches 12,16 --tune 6=16 2>&1 | grep -v amdgpu.ids
