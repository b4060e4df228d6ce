#!/bin/bash
# Round-5 evidence in one GPU call -> gpurun_out/r05/ (what is to be judged is copied into profiles/ as r05_*)
export ROUND=r05 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
T="--steps 8 --warmup 2 --no-cpu-baseline --no-sweep"
bash $R/tools/trace_bench.sh b64 $T
bash $R/tools/trace_bench.sh b8 $T --batch 8
bash $R/tools/trace_bench.sh b1 $T --batch 1
bash $R/tools/trace_bench.sh kv8_b64_ctx4096 $T --workload qwen2-7b-w4a16-kv8
bash $R/tools/trace_bench.sh w8a16_b16 $T --workload qwen2-7b-w8a16
bash $R/tools/trace_bench.sh tp2_shard_b64 $T --shard-of 2
bash $R/tools/trace_bench.sh llama70b_tp8_shard_b32 $T --workload llama3-70b-awq --shard-of 8
bash $R/tools/engine_traffic.sh > /dev/null 2>&1
cd $R
( python tools/attn_bench.py --product; python tools/attn_bench.py --product --ctx 4096 --int8; python tools/attn_bench.py --product --batch 16 ) 2>&1 | grep -v amdgpu.ids > $O/attn_bench.txt
python tools/batch_sweep.py 2>&1 | grep -v amdgpu.ids > $O/batch_sweep.txt
python tools/fullk64_time.py --product --ms 64,32,16,8 2>&1 | grep -v amdgpu.ids > $O/fullk64_time.txt
cp $O/traffic.json $R/profiles/r05_traffic.json 2>/dev/null   # so that the bench line below quotes the traffic of THESE sources
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/smoke.txt
export MI355_BENCH_ONE_GPU=1
for n in 2 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2973$n \
  bench.py --gpus $n --steps 8 --warmup 2 --no-sweep > $O/dryrun_${n}ranks_one_gpu.json 2> $O/dryrun_${n}ranks.log
done
ls -la $O
