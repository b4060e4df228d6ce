#!/bin/bash
# part C: attention micro-bench lines and the batch sweep 1..64
export ROUND=r03
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O; cd $R
( python tools/attn_bench.py; python tools/attn_bench.py --page 64; python tools/attn_bench.py --ctx 4096; python tools/attn_bench.py --ctx 4096 --int8; python tools/attn_bench.py --batch 16; python tools/attn_bench.py --batch 1 ) 2>&1 | grep -v amdgpu.ids > $O/attn_bench.txt
python tools/batch_sweep.py 2>&1 | grep -v amdgpu.ids > $O/batch_sweep.txt
cat $O/attn_bench.txt; tail -20 $O/batch_sweep.txt
