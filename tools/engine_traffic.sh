#!/bin/bash
# HBM traffic of the ENGINE's own launches: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace
# only) over `bench.py --no-graph` (eager launches of the C++ step: hipGraph replays are not broken down per kernel by the
# counter tool), summarised per kernel class by tools/engine_traffic.py.
# usage (on the GPU box): bash tools/engine_traffic.sh   -> gpurun_out/$ROUND/{pmc_engine_*.txt, traffic.json}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND:-r04}; mkdir -p $O
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd $R && timeout 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_engine_$ctr -o run -- \
      python bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $O/pmc_engine_$ctr.log 2>&1 )
done
# third pass, no counters: kernel durations of the same eager launches (rocprof begin -> end per dispatch, no launch gaps) for the roofline's
# kernel-time fraction (bench.py roofline.frac_kernel_time; VERDICT r05: the HIP-event time of eager launches includes the gaps)
( cd $R && timeout 240 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_engine_TIME -o run -- \
    python bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $O/pmc_engine_TIME.log 2>&1 )
python $R/tools/engine_traffic.py $O/pmc_engine_FETCH_SIZE $O/pmc_engine_WRITE_SIZE $O/pmc_engine_TIME > $O/pmc_engine_traffic.txt
cp $O/traffic.json $O/traffic.json.bak 2>/dev/null
rm -rf $O/pmc_engine_FETCH_SIZE $O/pmc_engine_WRITE_SIZE $O/pmc_engine_TIME
cat $O/pmc_engine_traffic.txt
