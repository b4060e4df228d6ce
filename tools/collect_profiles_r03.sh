#!/bin/bash
# Round-3 evidence in one GPU call -> gpurun_out/r03/ (copy what is to be judged into profiles/ as r03_*)
export ROUND=r03
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
bash $R/tools/trace_bench.sh b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
bash $R/tools/trace_bench.sh b1 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 1
bash $R/tools/engine_traffic.sh > /dev/null 2>&1
cd $R
( python tools/attn_bench.py; python tools/attn_bench.py --page 64; python tools/attn_bench.py --ctx 4096; python tools/attn_bench.py --ctx 4096 --int8; python tools/attn_bench.py --batch 16; python tools/attn_bench.py --batch 1 ) 2>&1 | grep -v amdgpu.ids > $O/attn_bench.txt
python tools/batch_sweep.py 2>&1 | grep -v amdgpu.ids > $O/batch_sweep.txt
cp $O/traffic.json $R/profiles/r03_traffic.json 2>/dev/null   # so that the bench line below quotes the traffic of THESE sources
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
ls -la $O
