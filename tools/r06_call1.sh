#!/bin/bash
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
bash tools/r06_ab_r04_vs_head.sh
( time python -m pytest tests/test_gpu_full_depth.py -m gpu -q -k "int8kv or tp2" 2>&1 | tail -30 ) > $O/new_full_depth_tests.txt 2>&1; cat $O/new_full_depth_tests.txt
cp gpurun_out/full_depth_parity.json $O/full_depth_parity_new.json
( MI355_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 8 --warmup 2 > $O/dryrun_selfspawn_2.json 2> $O/dryrun_selfspawn_2.log; echo rc=$? ) ; tail -c 1500 $O/dryrun_selfspawn_2.json; tail -5 $O/dryrun_selfspawn_2.log
