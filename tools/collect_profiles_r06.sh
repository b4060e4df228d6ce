#!/bin/bash
# Round-6 evidence in one GPU call -> gpurun_out/r06/ (what is to be judged is copied into profiles/ as r06_*)
export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
cd $R
# 1. the whole GPU suite on these sources
( time python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -8 ) > $O/full_gpu_suite.txt 2>&1; cat $O/full_gpu_suite.txt
cp gpurun_out/parity_greedy_ids.json $O/parity_greedy_ids.json; cp gpurun_out/full_depth_parity.json $O/full_depth_parity.json
# 2. kernel traces per configuration
T="--steps 8 --warmup 2 --no-cpu-baseline --no-sweep"
bash $R/tools/trace_bench.sh b64 $T
bash $R/tools/trace_bench.sh b8 $T --batch 8
bash $R/tools/trace_bench.sh b1 $T --batch 1
bash $R/tools/trace_bench.sh kv8_b64_ctx4096 $T --workload qwen2-7b-w4a16-kv8
bash $R/tools/trace_bench.sh w8a16_b16 $T --workload qwen2-7b-w8a16
bash $R/tools/trace_bench.sh tp2_shard_b64 $T --shard-of 2
bash $R/tools/trace_bench.sh tp4_shard_b64 $T --shard-of 4
bash $R/tools/trace_bench.sh llama70b_tp8_shard_b32 $T --workload llama3-70b-awq --shard-of 8
# 3. HBM traffic + kernel durations of the engine's own launches (three rocprofv3 passes), SQ counters
bash $R/tools/engine_traffic.sh > /dev/null 2>&1; cat $O/pmc_engine_traffic.txt
bash $R/tools/engine_pmc.sh > /dev/null 2>&1
cd $R
( python tools/attn_bench.py --product; python tools/attn_bench.py --product --ctx 4096 --int8; python tools/attn_bench.py --product --batch 16 ) 2>&1 | grep -v amdgpu.ids > $O/attn_bench.txt
python tools/batch_sweep.py 2>&1 | grep -v amdgpu.ids > $O/batch_sweep.txt
# 4. one rank's TP step
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'tok/s per rank-step', 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
( for so in 2 4; do python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b   one rank of tp$so b=64"; done
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b one rank of tp8 b=32"
  python bench.py --workload qwen2-72b-w4a16 --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "qwen2-72b  one rank of tp8 b=8 "
  python bench.py --workload qwen2-72b-w4a16 --shard-of 8 --batch 40 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "qwen2-72b  one rank of tp8 b=40"
  for so in 2 4; do for b in 1 8 16 32; do python bench.py --shard-of $so --batch $b --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b   one rank of tp$so b=$b"; done; done
) 2>&1 | tee $O/tp_shard_final.txt
# 5. the driver's line, the smoke, and the N = 2 / 8 launch lines with every rank on ONE GPU -- started by bench.py itself (no launcher)
cp $O/traffic.json $R/profiles/r06_traffic.json 2>/dev/null   # so that the bench line below quotes the traffic of THESE sources
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/smoke.txt
export MI355_BENCH_ONE_GPU=1
for n in 2 8; do
  timeout 1200 python bench.py --gpus $n --steps 8 --warmup 2 --no-sweep > $O/dryrun_selfspawn_${n}ranks_one_gpu.json 2> $O/dryrun_selfspawn_${n}ranks.log
  echo "n=$n rc=$?"; tail -1 $O/dryrun_selfspawn_${n}ranks_one_gpu.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['config']['parallelism'], d['value'], d['ms_per_step'], d.get('tp_layout',{}).get('error'), d['tp_layout'].get('ranks_bit_identical'), d['tp_layout'].get('hand_over'), d['roofline']['layout'], 'cpu_baseline' in d)"
done
rm -f $O/*.log.bak; ls -la $O | head -60
