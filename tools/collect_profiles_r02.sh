#!/bin/bash
# Round-2 evidence in one GPU call -> gpurun_out/r02/ (copy what is to be judged into profiles/ as r02_*):
#   kernel-trace summaries of bench.py at b = 64 and b = 1, the engine's own HBM traffic (FETCH_SIZE / WRITE_SIZE passes,
#   separate rocprofv3 runs), the un-profiled default bench line, attention / prefill-GEMM / speculative micro benches,
#   kernel durations of the fused full-K launches next to the launches they replace.
# usage (on the GPU box): bash tools/collect_profiles_r02.sh
export ROUND=r02
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
bash $R/tools/trace_bench.sh b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
bash $R/tools/trace_bench.sh b1 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 1
bash $R/tools/engine_traffic.sh > /dev/null 2>&1
cd $R
( python tools/attn_bench.py; python tools/attn_bench.py --ctx 4096; python tools/attn_bench.py --ctx 4096 --int8; python tools/attn_bench.py --batch 16 ) > $O/attn_bench.txt 2>&1
python tools/prefill_gemm_bench.py > $O/prefill_gemm_bench.txt 2>&1
python tools/spec_bench.py > $O/spec_round.txt 2>&1
bash tools/probe/ktrace.sh fullk python tools/fullk_bench.py --ms 1,8,64 > $O/fullk_kernel_durations.txt 2>&1
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
ls -la $O
