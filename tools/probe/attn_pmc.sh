#!/bin/bash
# SQ counters of the paged attention kernel (b = 64, ctx 1024 fp16 and ctx 4096 INT8), tuning-build micro-bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_pmc; mkdir -p $O; export TMPDIR=/tmp
for cfg in "fp16 --ctx 1024" "int8 --ctx 4096 --int8"; do
  tag=${cfg%% *}; args=${cfg#* }
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    i=$((i+1))
    ( cd $R && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_$i -o run -- python tools/attn_bench.py $args --iters 6 > $O/${tag}_$i.log 2>&1 )
    python - "$O/${tag}_$i" "$tag" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "paged_attn" not in k: continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(sys.argv[2], k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
  done
done
