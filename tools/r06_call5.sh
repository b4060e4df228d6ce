#!/bin/bash
# Round 6, VERDICT r05 item 6 (iii): s_setprio around the MFMA group of the wide GEMM's unit (0=32) / one wave of every SIMD pair at priority 2 (0=64) against the product stream (0=0):
# whole-step timing at b = 64 (alternating), kernel trace per variant, SQ counters per variant.
cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step', d['ms_per_step_repeats'], 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
R=$O/gemm_wide_setprio.txt; : > $R
for rep in 1 2; do for v in 0 32 64; do python bench.py --no-cpu-baseline --no-sweep --steps 64 --debug-set 0=$v 2>/dev/null | tail -1 | line "[0=$v] qwen2-7b tp1 b=64" >> $R; done; done
for v in 0 32 64; do
  bash tools/trace_bench.sh wide_prio_$v --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --debug-set 0=$v
  echo "== kernel trace, --debug-set 0=$v" >> $R; head -8 $O/kernel_stats_wide_prio_$v.txt >> $R
done
for v in 0 32 64; do BENCH_EXTRA="--debug-set 0=$v" PMC_TAG=_wide_prio_$v bash tools/engine_pmc.sh > /dev/null 2>&1; echo "== SQ counters, --debug-set 0=$v" >> $R; grep -i "gemm_wide\|kernel " $O/pmc_engine_sq_wide_prio_$v.txt | head -4 >> $R; done
cat $R
# the product library after the splitk64 fix: default line pieces + kernel trace
python bench.py --no-cpu-baseline --no-sweep --steps 64 --warmup 8 2>/dev/null | tail -1 | line "[product] qwen2-7b tp1 b=64"
bash tools/trace_bench.sh b64_mid --steps 8 --warmup 2 --no-cpu-baseline --no-sweep; head -9 $O/kernel_stats_b64_mid.txt
