#!/bin/bash
# Round 5, call G: the gate_up shard of the TP step as ONE launch from an image (gemm_splitk64 direct form): multi-process engine tests, one rank's TP step new vs
# staged split-K + fold (tuning build, switch 5 = 3).
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_allreduce.py -x -q -k "engine7b or engine-2 or engine70-8 or bf16-2" 2>&1 | tail -6
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'eager ms/step', d['step_roofline']['eager_kernel_ms_per_step'])"; }
( for so in 2 4; do
    python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b one rank of tp$so  gate_up shard in one launch"
    python bench.py --shard-of $so --no-cpu-baseline --no-sweep --steps 20 --debug-set 5=3 2>/dev/null | tail -1 | line "qwen2-7b one rank of tp$so  staged split-K + fold     "
  done
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 2>/dev/null | tail -1 | line "llama3-70b one rank of tp8 gate_up shard in one launch"
  python bench.py --workload llama3-70b-awq --shard-of 8 --no-cpu-baseline --no-sweep --steps 10 --debug-set 5=3 2>/dev/null | tail -1 | line "llama3-70b one rank of tp8 staged split-K + fold     "
  python bench.py --shard-of 2 --batch 8 --no-cpu-baseline --no-sweep --steps 20 2>/dev/null | tail -1 | line "qwen2-7b one rank of tp2 b=8 one launch"
  python bench.py --shard-of 2 --batch 8 --no-cpu-baseline --no-sweep --steps 20 --debug-set 5=3 2>/dev/null | tail -1 | line "qwen2-7b one rank of tp2 b=8 staged     "
) 2>&1 | tee $O/tp_gate_up_direct.txt
bash tools/trace_bench.sh tp2_shard_b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --shard-of 2
grep -v "at6native\|rocclr\|hipcub\|elementwise" $O/kernel_stats_tp2_shard_b64.txt | head -10 | cut -c1-170
