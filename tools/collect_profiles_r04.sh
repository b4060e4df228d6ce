#!/bin/bash
# Round-4 evidence in one GPU call -> gpurun_out/r04/ (copy what is to be judged into profiles/ as r04_*)
export ROUND=r04
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
bash $R/tools/trace_bench.sh b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
bash $R/tools/trace_bench.sh b8 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 8
bash $R/tools/trace_bench.sh b1 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 1
bash $R/tools/trace_bench.sh kv8_b64_ctx4096 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --workload qwen2-7b-w4a16-kv8
bash $R/tools/trace_bench.sh w8a16_b16 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --workload qwen2-7b-w8a16
bash $R/tools/engine_traffic.sh > /dev/null 2>&1
cd $R
( python tools/attn_bench.py --product; python tools/attn_bench.py --product --ctx 4096; python tools/attn_bench.py --product --ctx 4096 --int8; python tools/attn_bench.py --product --batch 16; python tools/attn_bench.py --product --copies 1 ) 2>&1 | grep -v amdgpu.ids > $O/attn_bench.txt
python tools/attn_stamps.py 2>&1 | grep -v amdgpu.ids > $O/attn_stamps.txt
python tools/batch_sweep.py 2>&1 | grep -v amdgpu.ids > $O/batch_sweep.txt
python tools/fullk64_time.py --product --ms 64,32,16,8 2>&1 | grep -v amdgpu.ids > $O/fullk64_time.txt
cp $O/traffic.json $R/profiles/r04_traffic.json 2>/dev/null   # so that the bench line below quotes the traffic of THESE sources
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
ls -la $O
