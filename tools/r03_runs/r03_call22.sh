#!/bin/bash
# dry runs of the driver's N = 2 / 4 launches of the DEFAULT workload on one GPU (all ranks on device 0): control plane, tp2 / tp4 engines
# of Qwen2-7B, IPC all-reduce inside the captured step, JSON line.  Timing meaningless.
mkdir -p gpurun_out/r03
export MI355_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2962$n \
  bench.py --gpus $n --steps 8 --warmup 2 --no-sweep --no-cpu-baseline > gpurun_out/r03/dryrun_${n}ranks.json 2> gpurun_out/r03/dryrun_${n}ranks.log
echo "n=$n rc=$?"
tail -1 gpurun_out/r03/dryrun_${n}ranks.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['config']['parallelism'], d['value'], d['scaling'], d.get('tp_layout',{}).get('error'), d.get('replica_layout',{}).get('tokens_per_s'))"
done
