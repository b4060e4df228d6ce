#!/usr/bin/env python3
"""Per-kernel means of SQ / GRBM counters from rocprofv3 --pmc csv directories (tools/engine_pmc.sh), for the hot kernels of the decode step.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs,
GRBM_GUI_ACTIVE cycles summed over the 8 XCDs (/opt/skills/guides/MI355X_MICROARCH.md).  Derived: MFMA pipe busy = MFMA_BUSY / (1024 SIMDs x
GUI_ACTIVE / 8); wait / stall / active = shares of SQ_WAVE_CYCLES.  usage: engine_pmc.py <dir> [<dir> ...]"""
import collections, csv, glob, re, sys
KERNELS = [("gemm_wide_kernel", "gate_up (gemm_wide)"), ("gemm_splitk64_kernel", "down (gemm_splitk64)"), (r"gemm_fullk64_kernel(ILi\d+ELi\d+ELi\d+ELi2E|<\d+, \d+, \d+, 2,)", "qkv+rope (gemm_fullk64)"),
           (r"gemm_fullk64_kernel(ILi\d+ELi\d+ELi\d+ELi1E|<\d+, \d+, \d+, 1,)", "o+residual (gemm_fullk64)"), ("paged_attn_kernel", "attention"), ("gemm_wq_kernel(ILi|<)16", "lm_head (gemm_wq W16)"), ("add_rmsnorm_kernel", "fold (add_rmsnorm)")]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            for pat, name in KERNELS:
                if re.search(pat, k):
                    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    break
print("# per launch means; b = 64 headline step, eager launches under rocprofv3 --pmc (two passes)")
for _, name in KERNELS:
    c = {k: sum(v) / len(v) for k, v in acc.get(name, {}).items()}
    if not c:
        continue
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    line = f"{name:28s} launches {len(next(iter(acc[name].values()))):5d}"
    if wc:
        line += f"  wave-cycles {wc / 1e6:7.2f} M quad  wait {100 * c.get('SQ_WAIT_ANY', 0) / wc:4.0f} %  issue-stall {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:4.0f} %  active {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:4.0f} %"
    if gui:
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        line += f"  | kernel {gui / 1e3:6.1f} k cycles/XCD  VALU insts {c.get('SQ_INSTS_VALU', 0) / 1e6:6.2f} M  MFMA ops(F16) {c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0) / 1e6:7.2f} M  MFMA pipe busy {100 * mf / 1024 / gui:5.1f} %"
    print(line)
