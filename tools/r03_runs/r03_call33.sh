#!/bin/bash
# full-K launches: helper publishes 1 / rms first (DPP reductions), merged meta loads: parity, then timing
timeout 600 python -m pytest tests/test_gpu_fused_linear.py -x -q 2>&1 | tail -4
timeout 400 python tools/fullk_time.py --sweep 5=0 --ms 1,2,4,8,16 2>&1 | grep -v amdgpu.ids
