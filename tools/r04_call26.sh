#!/bin/bash
cd $GRAFT_REPO_ROOT; export ROUND=r04; mkdir -p gpurun_out/r04
for v in "4=4" "3=12" "3=23" "3=40" "3=64"; do
  tag=pf_$(echo $v | tr '=' '_')
  bash tools/trace_bench.sh $tag --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --debug-set $v
  echo "== $v: $(python -c "import json; d=json.load(open('gpurun_out/r04/bench_under_rocprof_$tag.json')); print(d['ms_per_step'])")"
  grep -E "paged_attn_kernel|gemm_fullk64_kernelILi4ELi4ELi2|add_rmsnorm" gpurun_out/r04/kernel_stats_$tag.txt | awk '{print "   ", substr($1,1,60), $(NF-3), $(NF-2)}'
done 2>&1 | tee gpurun_out/r04_c26_kv_prefetch.txt
