#!/bin/bash
# A/B on one box: tuning lib = previous gemm_wide.hip (two-round merge), product lib = one-round reduce-scatter merge
for rep in 1 2 3; do
echo "OLD: $(python tools/gemm_bench.py --ms 64 --shapes gate_up --iters 200 2>&1 | tail -1)"
echo "NEW: $(python tools/gemm_bench.py --product --ms 64 --shapes gate_up --iters 200 2>&1 | tail -1)"
done
echo "OLD M=32,48: $(python tools/gemm_bench.py --ms 32,48 --shapes gate_up --iters 100 2>&1 | tail -2 | tr '\n' ' ')"
echo "NEW M=32,48: $(python tools/gemm_bench.py --product --ms 32,48 --shapes gate_up --iters 100 2>&1 | tail -2 | tr '\n' ' ')"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -x -q -k "wide or linear_baseline or silu" 2>&1 | tail -2
