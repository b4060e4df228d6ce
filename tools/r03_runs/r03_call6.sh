#!/bin/bash
# transport test + RCCL bootstrap probes
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_allreduce.py -x -q -m gpu -k "transport or engine-2" > gpurun_out/r03/pytest_transport.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/pytest_transport.txt
{
cat /proc/net/dev | head; hostname; hostname -i
for v in "" "NCCL_SOCKET_IFNAME=lo" "NCCL_SOCKET_FAMILY=AF_INET" "NCCL_COMM_ID=127.0.0.1:29777"; do
  for w in torch mine; do
    echo "=== env[$v] $w"
    env $v NCCL_DEBUG=WARN timeout 120 python tools/probe/rccl_probe2.py $w 2>&1 | grep -v "^$" | tail -12
  done
done
} > gpurun_out/r03/rccl_probe2.txt 2>&1
tail -5 gpurun_out/r03/pytest_transport.txt
