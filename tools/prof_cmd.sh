#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <command...>  -> gpurun_out/prof_<tag>/ (kernel trace), prints the per-(kernel, grid) summary
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace -d $out -o run -- "$@" > $out/cmd.log 2>&1 )
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls $out/*.db $out/*/*.db 2>/dev/null | head -1) --by-grid | head -${TOP:-14}
