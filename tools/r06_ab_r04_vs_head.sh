#!/bin/bash
# Round 6: the round-4-end library (73f9bd2, exported + built under ab_r04/) against HEAD's on ONE lease, alternating, plus a kernel trace of each.
# usage (GPU box): bash tools/r06_ab_r04_vs_head.sh   -> gpurun_out/r06/r04_vs_head_same_box.txt
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
R=$O/r04_vs_head_same_box.txt; : > $R
one() { # tag dir
  ( cd $2 && python bench.py --no-cpu-baseline --no-sweep --steps 64 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms_per_step', d['ms_per_step'], 'repeats', d['ms_per_step_repeats'], 'eager_ms', d['step_roofline']['eager_kernel_ms_per_step'])" ) >> $R 2>&1
}
for i in 1 2 3; do one r04 ab_r04; one head .; done
cat $R
for t in r04:ab_r04 head:.; do
  tag=${t%%:*}; dir=${t##*:}
  out=$GRAFT_REPO_ROOT/$O/prof_ab_$tag; mkdir -p $out
  ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/$dir && timeout 300 rocprofv3 --kernel-trace -d $out -o run -- python bench.py --no-cpu-baseline --no-sweep --steps 64 --warmup 8 > $out/cmd.log 2>&1 )
  echo "== kernel trace $tag" >> $R
  python tools/rocpd_summary.py $(ls $out/*.db $out/*/*.db 2>/dev/null | head -1) --by-grid | head -14 >> $R
  rm -rf $out      # the trace databases are tens of MB: gpurun copies back at most 64 MiB
done
tail -34 $R
