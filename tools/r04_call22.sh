#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 900 python -m pytest "tests/test_gpu_allreduce.py::test_custom_allreduce_processes_on_one_gpu" -x -q -k "engine70full" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 ) 2>&1 | tee gpurun_out/r04_c22_engine70full.txt
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "per_rank" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | tee gpurun_out/r04_c22_per_rank.txt
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest "tests/test_gpu_allreduce.py::test_custom_allreduce_processes_on_one_gpu" -x -q -k "engine-2" 2>&1 | grep -E "passed|failed" ; done | sort | uniq -c | tee gpurun_out/r04_c22_engine2_alone_x10.txt
( time timeout 1200 python -m pytest tests/test_gpu_allreduce.py -q -n 4 2>&1 | grep -E "passed|failed|FAILED" | tail -5 ) 2>&1 | tee gpurun_out/r04_c22_allreduce_xdist4.txt
