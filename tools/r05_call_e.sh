#!/bin/bash
# Round 5, call E: SQ counters of the headline step's kernels, configs[1] at full depth, the INT8-KV batch sweep, one rank's TP step over the batch sizes.
cd $GRAFT_REPO_ROOT; export ROUND=r05 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_full_depth.py -x -q -k w8a16 2>&1 | tail -4
cp gpurun_out/full_depth_parity.json $O/full_depth_parity_w8.json
bash tools/engine_pmc.sh 2>&1 | tail -12
python tools/batch_sweep.py --kv8 --ctx 4096 --batches 1,8,16,32,64 2>&1 | grep -v amdgpu.ids > $O/batch_sweep_kv8_ctx4096.txt; cat $O/batch_sweep_kv8_ctx4096.txt
( for so in 2 4; do for b in 1 8 16 32 64; do
    python bench.py --shard-of $so --batch $b --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('qwen2-7b one rank of tp$so  b=$b', d['ms_per_step'], 'ms/step  p50', d['p50_ms'])"
  done; done ) 2>&1 | tee $O/tp_shard_batch_sweep.txt
bash tools/trace_bench.sh tp2_shard_b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --shard-of 2
bash tools/trace_bench.sh llama70b_tp8_shard_b32 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --workload llama3-70b-awq --shard-of 8
