#!/bin/bash
# Round 5, call F: under TP, from how many rows do the image launches (QKV shard, down shard as K quarters) beat the few-row QKV launch + staged down?
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
( for so in 2 4; do for b in 1 2 4; do
    python bench.py --shard-of $so --batch $b --no-cpu-baseline --no-sweep --steps 20 --debug-set 6=4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one rank of tp$so  b=$b  few-row QKV + staged down up to 4 rows', d['ms_per_step'])"
    python bench.py --shard-of $so --batch $b --no-cpu-baseline --no-sweep --steps 20 --debug-set 6=99 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one rank of tp$so  b=$b  image launches from 1 row              ', d['ms_per_step'])"
  done; done ) 2>&1 | tee $O/tp_small_batch_crossover.txt
