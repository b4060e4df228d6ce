#!/bin/bash
# SQ counters of the M = 64 gate_up GEMM (tuning build): where do the wave cycles go?  usage: bash tools/probe/pc_pmc.sh "5=0" tag
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pc_pmc; mkdir -p $O; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  ( cd $R && MI355_TUNING_LIB=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$2_$i -o run -- \
      python tools/gemm_bench.py --ms 64 --shapes gate_up --iters 4 --copies 2 --tune $1 > $O/$2_$i.log 2>&1 )
  python - "$O/$2_$i" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
done
