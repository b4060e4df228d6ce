#!/bin/bash
# full-K launches: interleaved K slices + norm partial sums requested first (product) against the previous source (tuning library), one box
timeout 400 python -m pytest tests/test_gpu_fused_linear.py -x -q 2>&1 | tail -3
timeout 300 python tools/fullk_time.py --ms 1,4,8,16 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/fullk_time.py --product --ms 1,4,8,16 2>&1 | grep -v amdgpu.ids
