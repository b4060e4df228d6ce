#!/bin/bash
# full-K launches with the helper wave + dense activation loads: parity, then timing (tuning build, same source: 5=1 forces the fragment-shaped loads)
timeout 600 python -m pytest tests/test_gpu_fused_linear.py -x -q 2>&1 | tail -4
timeout 400 python tools/fullk_time.py --sweep 5=1,0 --ms 1,4,8,16 2>&1 | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "small" 2>&1 | tail -2
