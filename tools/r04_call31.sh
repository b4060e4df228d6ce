#!/bin/bash
# smoke() as the driver runs it, then dry runs of the driver's N = 2 / 8 launch lines with all ranks on ONE GPU (control plane, TP engines,
# IPC all-reduce inside the captured step, JSON line incl. ms_per_step_repeats of the TP layout).  Timing meaningless.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/smoke.txt
export MI355_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2963$n \
  bench.py --gpus $n --steps 8 --warmup 2 --no-sweep --no-cpu-baseline > gpurun_out/r04/dryrun_${n}ranks.json 2> gpurun_out/r04/dryrun_${n}ranks.log
echo "n=$n rc=$?"
tail -1 gpurun_out/r04/dryrun_${n}ranks.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['config']['parallelism'], d['value'], d['ms_per_step'], d['ms_per_step_repeats'], d['scaling'], d.get('tp_layout',{}).get('error'), d.get('replica_layout',{}).get('ms_per_step_repeats'))"
done
