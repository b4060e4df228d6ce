#!/bin/bash
# Round 6: the closed experiments, reproducible on one GPU (tuning build: python -m rtp_llm_amd.build --tuning).  usage (GPU box): bash tools/r06_experiments.sh [ring|setprio|phase]
#   ring     deeper activation rings of the full-K image launches at <= 32 rows per block (--debug-set 9=1: 4 MB fragments in flight, 9=2: 8) -> profiles/r06_fullk64_ring_depth.txt
#   phase    the second wave of every SIMD enters the wide GEMM's loop half a unit (0=128) / a whole unit (0=256) behind its partner -> profiles/r06_gemm_wide_phase_offset.txt
#   setprio  the wide GEMM's unit with s_setprio over its MFMA group (0=32) / one wave of every SIMD pair at priority 2 (0=64): step time, kernel trace, SQ counters -> profiles/r06_gemm_wide_setprio.txt
# (the in-launch attention merge needs tools/experiments/r06_attention_inlaunch_merge_tickets.patch applied; its switch is --debug-set 10=1 = reduce launch, 10=0 = merged)
cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step', d['ms_per_step_repeats'], 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
what=${1:-ring}
if [ $what = ring ]; then
  R=$O/fullk64_ring_depth.txt; : > $R
  for v in 0 1 2; do
    for b in 8 16 32 64; do python bench.py --batch $b --no-cpu-baseline --no-sweep --steps 20 --debug-set 9=$v 2>/dev/null | tail -1 | line "[9=$v] qwen2-7b tp1 b=$b" >> $R; done
    python bench.py --shard-of 2 --no-cpu-baseline --no-sweep --steps 20 --debug-set 9=$v 2>/dev/null | tail -1 | line "[9=$v] qwen2-7b one rank of tp2 b=64" >> $R
    python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 --debug-set 9=$v 2>/dev/null | tail -1 | line "[9=$v] llama3-70b one rank of tp8 b=32" >> $R
  done
  cat $R
else
  if [ $what = phase ]; then VARS="0 128 256"; R=$O/gemm_wide_phase_offset.txt; else VARS="0 32 64"; R=$O/gemm_wide_setprio.txt; fi; : > $R
  for rep in 1 2; do for v in $VARS; do python bench.py --no-cpu-baseline --no-sweep --steps 64 --debug-set 0=$v 2>/dev/null | tail -1 | line "[0=$v] qwen2-7b tp1 b=64" >> $R; done; done
  for v in $VARS; do
    bash tools/trace_bench.sh wide_prio_$v --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --debug-set 0=$v
    echo "== kernel trace, --debug-set 0=$v" >> $R; head -8 $O/kernel_stats_wide_prio_$v.txt >> $R
  done
  for v in $VARS; do BENCH_EXTRA="--debug-set 0=$v" PMC_TAG=_wide_prio_$v bash tools/engine_pmc.sh > /dev/null 2>&1; echo "== SQ counters, --debug-set 0=$v" >> $R; grep -i "gemm_wide\|kernel " $O/pmc_engine_sq_wide_prio_$v.txt | head -4 >> $R; done
  cat $R
fi
