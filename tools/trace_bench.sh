#!/bin/bash
# Per-kernel timing of bench.py under rocprofv3 (kernel trace only).  usage (on the GPU box): bash tools/trace_bench.sh TAG [bench args...]
# Output: gpurun_out/$ROUND/kernel_stats_TAG.txt + bench_under_rocprof_TAG.json   (ROUND defaults to r02)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND:-r03}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o run -- python $R/bench.py "$@" > $O/bench_under_rocprof_$tag.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof_$tag.log | tail -1 > $O/bench_under_rocprof_$tag.json
python $R/tools/rocpd_summary.py $(ls $O/trace_$tag/*.db $O/trace_$tag/*/*.db 2>/dev/null | head -1) --by-grid > $O/kernel_stats_$tag.txt
rm -rf $O/trace_$tag
