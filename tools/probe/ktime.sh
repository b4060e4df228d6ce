#!/bin/bash
# Kernel-trace durations of a gemm_bench run (host-loop timing floors at ~33 us per ctypes call).  usage: ktime.sh tag <gemm_bench args>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ktime/$1; shift; mkdir -p $O; export TMPDIR=/tmp
( cd $R && MI355_TUNING_LIB=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python tools/gemm_bench.py "$@" > $O/log.txt 2>&1 )
python - "$O" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] or "reduce" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:70], r["Grid_Size_X"] if "Grid_Size_X" in r else "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v); n = len(v)
    print(f"{k[0]:72s} grid {k[1]:>8s} n={n:4d} med {v[n // 2]:7.2f} us  min {v[0]:7.2f}")
PY
