#!/bin/bash
cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step', 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
# 1. deeper activation rings of the full-K image launches at <= 32 rows per block (tuning build, --debug-set 9=1: 4 MB fragments, 9=2: 8)
R=$O/fullk64_ring_depth.txt; : > $R
for v in 0 1 2; do
  for b in 8 16 32 64; do python bench.py --batch $b --no-cpu-baseline --no-sweep --steps 20 --debug-set 9=$v 2>/dev/null | tail -1 | line "[9=$v] qwen2-7b tp1 b=$b" >> $R; done
  python bench.py --shard-of 2 --no-cpu-baseline --no-sweep --steps 20 --debug-set 9=$v 2>/dev/null | tail -1 | line "[9=$v] qwen2-7b one rank of tp2 b=64" >> $R
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 --debug-set 9=$v 2>/dev/null | tail -1 | line "[9=$v] llama3-70b one rank of tp8 b=32" >> $R
done
cat $R
# 2. kernel traces of one rank's step with the published O / down shards
bash tools/trace_bench.sh llama70b_tp8_shard_b32 --workload llama3-70b-awq --shard-of 8 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
head -14 $O/kernel_stats_llama70b_tp8_shard_b32.txt
bash tools/trace_bench.sh tp2_shard_b64 --shard-of 2 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
head -14 $O/kernel_stats_tp2_shard_b64.txt
# 2b. partitions merged inside the attention launch (--debug-set 10=1: the reduce launch of rounds 1-5)
R2=$O/attn_merge_in_launch.txt; : > $R2
for v in 1 0; do
  for b in 1 4 8 16 32; do python bench.py --batch $b --no-cpu-baseline --no-sweep --steps 20 --debug-set 10=$v 2>/dev/null | tail -1 | line "[10=$v] qwen2-7b tp1 b=$b" >> $R2; done
  python bench.py --workload qwen2-7b-w4a16-kv8 --batch 16 --no-cpu-baseline --no-sweep --steps 20 --debug-set 10=$v 2>/dev/null | tail -1 | line "[10=$v] qwen2-7b kv8 ctx4096 b=16" >> $R2
  python bench.py --shard-of 2 --no-cpu-baseline --no-sweep --steps 20 --debug-set 10=$v 2>/dev/null | tail -1 | line "[10=$v] qwen2-7b one rank of tp2 b=64" >> $R2
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 --debug-set 10=$v 2>/dev/null | tail -1 | line "[10=$v] llama3-70b one rank of tp8 b=32" >> $R2
done
cat $R2
( time python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -12
# 3. the multi-process tests again (startup check bounded on a shared device)
( time python -m pytest tests/test_gpu_allreduce.py tests/test_gpu_tp_engine.py -m gpu -q 2>&1 | tail -8 ) > $O/tp_tests.txt 2>&1; grep -v "^\[W\|amdgpu.ids\|Gloo" $O/tp_tests.txt | tail -12
