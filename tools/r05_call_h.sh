#!/bin/bash
# Round 5, call H: attention with 128 (sequence, kv head) pairs -- two 512-token partitions + the reduce launch (the planner's choice) or ONE partition per pair, in the graph-replayed step.
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'attn eager ms/step', d['step_roofline']['eager_kernel_ms_per_step']['attn'])"; }
( python bench.py --batch 32 --no-cpu-baseline --no-sweep --steps 30 --debug-set 0=0 2>/dev/null | tail -1 | line "tp1 b=32 planner (2 x 512 + reduce)"
  python bench.py --batch 32 --no-cpu-baseline --no-sweep --steps 30 --attn-ps 1024 2>/dev/null | tail -1 | line "tp1 b=32 one partition per pair   "
  python bench.py --batch 48 --no-cpu-baseline --no-sweep --steps 30 --debug-set 0=0 2>/dev/null | tail -1 | line "tp1 b=48 planner                   "
  python bench.py --batch 48 --no-cpu-baseline --no-sweep --steps 30 --attn-ps 1024 2>/dev/null | tail -1 | line "tp1 b=48 one partition per pair   "
  python bench.py --batch 16 --no-cpu-baseline --no-sweep --steps 30 --debug-set 0=0 2>/dev/null | tail -1 | line "tp1 b=16 planner (4 x 256 + reduce)"
  python bench.py --batch 16 --no-cpu-baseline --no-sweep --steps 30 --attn-ps 512 2>/dev/null | tail -1 | line "tp1 b=16 2 x 512 + reduce          "
  python bench.py --shard-of 2 --no-cpu-baseline --no-sweep --steps 30 --debug-set 0=0 2>/dev/null | tail -1 | line "one rank of tp2 b=64 planner       "
  python bench.py --shard-of 2 --no-cpu-baseline --no-sweep --steps 30 --attn-ps 1024 2>/dev/null | tail -1 | line "one rank of tp2 b=64 one partition "
) 2>&1 | tee $O/attn_partitions_in_step.txt
