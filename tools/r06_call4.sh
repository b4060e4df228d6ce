#!/bin/bash
cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
( time python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -40 ) > $O/full_gpu_suite.txt 2>&1; tail -45 $O/full_gpu_suite.txt
