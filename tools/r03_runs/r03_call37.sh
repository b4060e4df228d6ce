#!/bin/bash
# paged attention with the partition merge inside the launch: engine parity, then the step with (default) and without (7=1) it
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_speculative.py tests/test_gpu_checkpoint.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "engine or graph or step" 2>&1 | tail -3
timeout 400 python tools/batch_sweep.py --batches 1,2,4,8,12,16,32 --tune 7=1 2>&1 | grep -v amdgpu.ids
timeout 400 python tools/batch_sweep.py --batches 1,2,4,8,12,16,32 --tune 7=0 2>&1 | grep -v amdgpu.ids
