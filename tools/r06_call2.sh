#!/bin/bash
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
( time python -m pytest tests/test_gpu_allreduce.py tests/test_gpu_tp_engine.py -m gpu -x -q 2>&1 | tail -25 ) > $O/tp_tests.txt 2>&1; cat $O/tp_tests.txt
( time python -m pytest tests/test_gpu_full_depth.py -m gpu -x -q -k "tp2" 2>&1 | tail -25 ) > $O/tp2_full_depth.txt 2>&1; cat $O/tp2_full_depth.txt
cp gpurun_out/full_depth_parity.json $O/full_depth_parity_tp2.json
bash tools/r06_tp_publish_ab.sh
