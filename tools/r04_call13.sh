#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python tools/batch_sweep.py --batches 12,13,14,16,17,20,24,32,40,48,64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c13_sweep_new.txt
timeout 400 python tools/batch_sweep.py --batches 13,16,17,24,32 --tune 5=2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c13_sweep_old.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "engine_wide_hidden_mid_batch or multi_row or greedy" 2>&1 | tail -3
