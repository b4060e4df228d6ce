#!/bin/bash
# Round 5, call I: attention with EIGHT waves per block (two per SIMD, NG = 2 groups in flight each; 253-254 registers, no scratch since the round-4 rework) vs the shipped four.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
( for args in "--ctx 4096 --int8" "--ctx 1024 --int8" "--ctx 1024" "--ctx 4096" "--ctx 1024 --batch 16" "--ctx 1024 --batch 8 --int8"; do
    python tools/attn_bench.py $args --tune 6=0 --iters 100 2>&1 | grep "^attn" | sed "s/^/4 waves (shipped)  /"
    python tools/attn_bench.py $args --tune 6=8 --iters 100 2>&1 | grep "^attn\|tune" | sed "s/^/8 waves, NG = 2    /"
  done ) 2>&1 | tee $O/attn_eight_waves.txt
