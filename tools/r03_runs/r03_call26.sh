#!/bin/bash
# A/B on one box: tuning lib = previous attention.hip, product lib = speculative first block-table lookup
for rep in 1 2; do for args in "" "--batch 16" "--batch 1" "--ctx 4096 --int8"; do
echo "OLD $args: $(python tools/attn_bench.py $args --iters 200 2>&1 | tail -1)"
echo "NEW $args: $(python tools/attn_bench.py --product $args --iters 200 2>&1 | tail -1)"
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -k "attention or engine_greedy or ragged or multi_row" 2>&1 | tail -2
