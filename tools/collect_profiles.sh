#!/bin/bash
# Round evidence, one GPU call: kernel-trace summaries of bench.py (b=64, b=1), PMC passes on the GEMM microbench
# (HBM read / write bytes per launch; SQ counters of the wide-M kernel), probes.  Output: gpurun_out/r01/ (copy what is
# to be judged into profiles/).  usage (on the GPU box): bash tools/collect_profiles.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() { # tag, bench args...
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o run -- python $R/bench.py "$@" > $O/bench_under_rocprof_$tag.log 2>&1
  tail -1 $O/bench_under_rocprof_$tag.log > $O/bench_under_rocprof_$tag.json
  python $R/tools/rocpd_summary.py $(ls $O/trace_$tag/*.db $O/trace_$tag/*/*.db 2>/dev/null | head -1) --by-grid > $O/kernel_stats_$tag.txt
  rm -rf $O/trace_$tag
}
trace b64 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep
trace b1 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --batch 1
pmc() { # tag, counters, cmd...
  tag=$1; ctr=$2; shift; shift
  ( cd $R && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$tag -o run -- "$@" > $O/pmc_$tag.log 2>&1 )
  python $R/tools/pmc_sum.py $O/pmc_$tag gemm > $O/pmc_$tag.txt
  rm -rf $O/pmc_$tag
}
for sh in qkv o down; do
  pmc fetch_$sh FETCH_SIZE python tools/gemm_bench.py --ms 64 --partial 1 --shapes $sh --iters 12
  pmc write_$sh WRITE_SIZE python tools/gemm_bench.py --ms 64 --partial 1 --shapes $sh --iters 12
done
pmc fetch_gate_up FETCH_SIZE python tools/gemm_bench.py --ms 64 --shapes gate_up --iters 12
pmc write_gate_up WRITE_SIZE python tools/gemm_bench.py --ms 64 --shapes gate_up --iters 12
pmc sq_wide "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU" python tools/gemm_bench.py --ms 64 --shapes gate_up --iters 12
pmc sq_wide2 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" python tools/gemm_bench.py --ms 64 --shapes gate_up --iters 12
( cd $R && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result -o tools/probe/unit_rate tools/probe/unit_rate.hip && ./tools/probe/unit_rate > $O/probe_unit_rate.txt 2>&1; python tools/wide_stamps.py 3584 > $O/wide_stamps.txt 2>&1; python tools/spec_bench.py > $O/spec_round.txt 2>&1 )
( cd $R && python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json )
ls -la $O
