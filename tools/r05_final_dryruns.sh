#!/bin/bash
# dry runs of the driver's N = 2 / 8 launch lines with every rank on ONE GPU, final sources
cd $GRAFT_REPO_ROOT; export MI355_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
for n in 2 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2983$n \
  bench.py --gpus $n --steps 8 --warmup 2 --no-sweep > $O/dryrun_${n}ranks_one_gpu.json 2> $O/dryrun_${n}ranks.log
echo "n=$n rc=$?"; tail -1 $O/dryrun_${n}ranks_one_gpu.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['config']['parallelism'], d['value'], d['ms_per_step'], d.get('tp_layout',{}).get('error'), d['tp_layout'].get('ranks_bit_identical'), d['tp_layout'].get('hand_over'), d['roofline']['layout'], 'cpu_baseline' in d)"
done
