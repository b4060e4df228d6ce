#!/bin/bash
# Round-3 GPU call 1: full GPU suite, attention micro-bench (NG = 2 / 3, page 16 / 64, fp16 / INT8), bench line, kernel trace at b = 64.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_call1.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_call1.txt; tail -5 $O/pytest_call1.txt
( python tools/attn_bench.py; python tools/attn_bench.py --tune 6=3; python tools/attn_bench.py --page 64; python tools/attn_bench.py --page 64 --tune 6=3;
  python tools/attn_bench.py --ctx 4096; python tools/attn_bench.py --ctx 4096 --tune 6=3; python tools/attn_bench.py --ctx 4096 --int8; python tools/attn_bench.py --ctx 4096 --int8 --tune 6=3;
  python tools/attn_bench.py --batch 16; python tools/attn_bench.py --batch 1; python tools/attn_bench.py --batch 8 ) > $O/attn_bench_call1.txt 2>&1
cat $O/attn_bench_call1.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_call1.log 2>&1; tail -1 $O/bench_call1.log > $O/bench_call1.json; cat $O/bench_call1.json
ROUND=r03 bash tools/trace_bench.sh b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
head -30 $O/kernel_stats_b64.txt
