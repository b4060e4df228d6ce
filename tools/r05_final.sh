#!/bin/bash
# Round 5, final call: the whole GPU suite on the final sources, then the evidence collection (tools/collect_profiles_r05.sh + tools/engine_pmc.sh).
cd $GRAFT_REPO_ROOT; export ROUND=r05 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r05; mkdir -p $O
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/full_gpu_suite.txt 2>&1; cat $O/full_gpu_suite.txt
cp gpurun_out/parity_greedy_ids.json $O/parity_greedy_ids.json; cp gpurun_out/full_depth_parity.json $O/full_depth_parity.json
bash tools/collect_profiles_r05.sh 2>&1 | tail -5
bash tools/engine_pmc.sh 2>&1 | tail -9
