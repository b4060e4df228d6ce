#!/bin/bash
# full-K launches (LDS-qualified poll, helper's requests ahead of the K waves' through a barrier without the vmcnt drain): parity, kernels, step
timeout 600 python -m pytest tests/test_gpu_fused_linear.py -x -q 2>&1 | tail -3
timeout 400 python tools/fullk_time.py --sweep 5=0 --ms 1,2,4,8,16 2>&1 | grep -v amdgpu.ids
timeout 400 python tools/batch_sweep.py --batches 1,2,4,8,9,10,12,16 --tune 6=10 2>&1 | grep -v amdgpu.ids
timeout 400 python tools/batch_sweep.py --batches 12,16 --tune 6=16 2>&1 | grep -v amdgpu.ids
