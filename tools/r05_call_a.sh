#!/bin/bash
# Round 5, call A: (1) the bf16 > 65504 image test, (2) attention partition-size / groups-in-flight sweep at small batches (tuning build),
# (3) dry runs of the driver's N = 2 / 8 launch lines with all ranks on ONE GPU (the line must carry roofline + cpu_baseline).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_bf16.py -x -q -k beyond 2>&1 | tail -12
( for b in 1 4 8 16 32; do for ps in 0 256 512 1024; do for ng in 2 3; do
    python tools/attn_bench.py --batch $b --ps $ps --tune 6=$ng --iters 200 2>&1 | grep "^attn" | sed "s/^/NG=$ng /"
  done; done; done ) > $O/attn_ps_sweep.txt 2>&1
cat $O/attn_ps_sweep.txt
export MI355_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2963$n \
  bench.py --gpus $n --steps 8 --warmup 2 --no-sweep > $O/dryrun_${n}ranks_one_gpu.json 2> $O/dryrun_${n}ranks.log
echo "n=$n rc=$?"
tail -1 $O/dryrun_${n}ranks_one_gpu.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['config']['parallelism'], d['value'], d['ms_per_step'], d['scaling'], d.get('tp_layout',{}).get('error'), 'roofline' in d, d.get('roofline',{}).get('layout'), d.get('roofline',{}).get('frac'), 'cpu_baseline' in d)"
done
