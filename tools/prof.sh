#!/bin/bash
# usage: tools_prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/ (kernel trace + stats)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats -d $out -o run -- python $GRAFT_REPO_ROOT/bench.py "$@" > $out/bench.log 2>&1
echo "rocprof rc=$?"
ls -R $out | head -30
