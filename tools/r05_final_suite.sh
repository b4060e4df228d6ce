#!/bin/bash
# the whole GPU suite on the final sources -> gpurun_out/r05/{full_gpu_suite.txt, parity_greedy_ids.json, full_depth_parity.json}
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/full_gpu_suite.txt 2>&1; cat $O/full_gpu_suite.txt
cp gpurun_out/parity_greedy_ids.json $O/parity_greedy_ids.json; cp gpurun_out/full_depth_parity.json $O/full_depth_parity.json
