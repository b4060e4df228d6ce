#!/bin/bash
# the partition merge as ONE memory round trip: attention tests, per-kernel time at b = 1 / 8 (rocprofv3), batch sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention or attn or engine_greedy" 2>&1 | tail -4
export TMPDIR=/tmp
for b in 1 8; do
  rm -rf /tmp/prof_b$b
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$b -o run -- python bench.py --batch $b --no-sweep --no-cpu-baseline --steps 30 --warmup 5 > /dev/null 2>&1
  f=$(find /tmp/prof_b$b -name "*kernel_stats.csv" | head -1)
  echo "== b=$b"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('attn','add_rmsnorm','gemm_fullk','gemm_wide','gemm_splitk')):
        print(f"{n[:70]:70s} n={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f}")
PY
done 2>&1 | tee gpurun_out/r04/attn_reduce_one_round_trip.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_reduce.json 2> gpurun_out/r04/bench_reduce.log; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_reduce.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [(s['batch'], s['ms_per_step']) for s in d['sweep']])
PY
