cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step', d['ms_per_step_repeats'], 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
R=$O/unit_9valu_timing_only.txt; : > $R
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-sweep --steps 64 --debug-set 15=0 2>/dev/null | tail -1 | line "[product stream: 13 VALU + 4 MFMA] qwen2-7b tp1 b=64" >> $R
  python bench.py --no-cpu-baseline --no-sweep --steps 64 --debug-set 0=512,7=9 2>/dev/null | tail -1 | line "[timing only: 9 VALU + 4 MFMA]  qwen2-7b tp1 b=64" >> $R
done
bash tools/trace_bench.sh unit13 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --debug-set 15=0; echo "== kernel trace, product stream" >> $R; head -6 $O/kernel_stats_unit13.txt | cut -c1-150 >> $R
bash tools/trace_bench.sh unit9 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --debug-set 0=512,7=9; echo "== kernel trace, timing-only 9-VALU unit (gate_up + down)" >> $R; head -6 $O/kernel_stats_unit9.txt | cut -c1-150 >> $R
cat $R
