#!/bin/bash
# Kernel-trace durations of any command, grouped by (kernel, grid).  usage: ktrace.sh tag <command...>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ktrace/$1; shift; mkdir -p $O; export TMPDIR=/tmp
( cd $R && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- "$@" > $O/log.txt 2>&1 )
python - "$O" <<'PY'
import csv, glob, sys, collections
d = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocclr" in n: continue
        k = (n.replace("(anonymous namespace)::", "").replace("void ", "")[:64], r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", ""))
        d.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v); n = len(v)
    print(f"{k[0]:66s} grid {k[1]:>8s} wg {k[2]:>5s} n={n:4d} med {v[n // 2]:7.2f} us  min {v[0]:7.2f}")
PY
