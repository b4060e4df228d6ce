#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <command...>  -> gpurun_out/pmc_<tag>/
tag=$1; shift; ctrs=$1; shift
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o run -- "$@" > $out/cmd.log 2>&1
echo "pmc rc=$?"; ls $out | head
