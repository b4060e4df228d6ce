#!/bin/bash
# dry run of the driver's N=8 launch on ONE GPU (8 ranks share device 0): control plane, TP=8 engine with Llama-3-70B's per-rank
# head shape (8 q / 1 kv head), IPC all-reduce at world 8 inside the captured step.  Timing is meaningless (8 ranks time-slice one GPU).
mkdir -p gpurun_out/r03
export MI355_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 8 --workload llama3-70b-awq --layers 2 --steps 8 --warmup 2 --no-sweep --no-cpu-baseline \
  > gpurun_out/r03/dryrun_8ranks_one_gpu.json 2> gpurun_out/r03/dryrun_8ranks_one_gpu.log
echo "rc=$?" >> gpurun_out/r03/dryrun_8ranks_one_gpu.log
tail -c 3000 gpurun_out/r03/dryrun_8ranks_one_gpu.json; tail -20 gpurun_out/r03/dryrun_8ranks_one_gpu.log
