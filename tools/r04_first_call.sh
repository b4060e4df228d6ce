#!/bin/bash
# First GPU call of the next round (~2 GPU-minutes): what the last round could not re-check after its GPU minutes ran out, and the
# probe its conclusions want.
#  1. the one GPU case that failed under 4 xdist workers at the end of round 3, alone
#  2. cost of a vector-memory wave-instruction per CU by kind + L2 -> CU fill rate (tools/probe/vmem_rate.hip, built by the caller:
#     cd tools/probe && hipcc -O3 --offload-arch=gfx950 vmem_rate.hip -o vmem_rate)
#  3. the few-row full-K launches as committed (tools/fullk_time.py, product library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python -m pytest "tests/test_gpu_allreduce.py::test_custom_allreduce_processes_on_one_gpu" -x -q -k "engine-2 or kernels-2" 2>&1 | tail -5
[ -x tools/probe/vmem_rate ] && timeout 120 tools/probe/vmem_rate 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_vmem_rate.txt
timeout 300 python tools/fullk_time.py --product --ms 1,4,8,12,16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_fullk_time.txt
