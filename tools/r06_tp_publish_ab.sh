#!/bin/bash
# Round 6: one rank's TP step (bench.py --shard-of N, world-1 context) with the O / down shards PUBLISHED by their full-K GEMM into the registered
# all-reduce buffer (mi355_linear_publish_img + mi355_allreduce_fused_published_dt) against the round-5 chain (slabs -> fold + publish inside the all-reduce
# launch).  Tuning build: --debug-set 8=3 switches both off, 8=1 O off, 8=2 down off, 8=4 also takes the down shard past K 5760.
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step', 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
R=$O/tp_publish_ab.txt; : > $R
for v in 3 0 1 2 4; do
  for so in 2 4; do python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 --debug-set 8=$v 2>/dev/null | tail -1 | line "[8=$v] qwen2-7b   one rank of tp$so b=64" >> $R; done
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 --debug-set 8=$v 2>/dev/null | tail -1 | line "[8=$v] llama3-70b one rank of tp8 b=32" >> $R
done
for so in 2 4; do for b in 1 8; do python bench.py --shard-of $so --batch $b --no-cpu-baseline --no-sweep --steps 20 --debug-set 8=3 2>/dev/null | tail -1 | line "[8=3] qwen2-7b   one rank of tp$so b=$b" >> $R
  python bench.py --shard-of $so --batch $b --no-cpu-baseline --no-sweep --steps 20 --debug-set 8=0 2>/dev/null | tail -1 | line "[8=0] qwen2-7b   one rank of tp$so b=$b" >> $R; done; done
cat $R
