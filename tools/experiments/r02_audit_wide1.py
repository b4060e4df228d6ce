#!/usr/bin/env python3
"""ISA audit of gemm_wide1: inside the main loop, no compiler-emitted instruction (anything outside #ASMSTART/#ASMEND) may WRITE
or COPY a register of the fixed map while loads issued from asm can be in flight -- reads of ring registers right after their wait
(v_perm of the meta word, the dword handed to the unit) are the only legitimate touches.  usage: audit_wide1.py file.s"""
import re, sys
src = open(sys.argv[1]).read()
m = re.search(r"^_ZN12_GLOBAL__N_117gemm_wide1_kernel.*?:\n(.*?)s_endpgm", src, re.S | re.M)
lines = m.group(1).split("\n")
hdr = next(i for i, l in enumerate(lines) if "Loop Header" in l)
label = lines[hdr].split(":")[0].strip()
end = max(i for i, l in enumerate(lines) if ("s_cbranch" in l or "s_branch" in l) and label in l)   # the back edge
pinned_v = set(range(80, 252)); pinned_a = set(range(0, 224))
def regs(tok):
    out = []
    for mm in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        out += [(mm.group(1), r) for r in range(int(mm.group(2)), int(mm.group(3)) + 1)] if mm.group(1) else [(mm.group(4), int(mm.group(5)))]
    return out
# two trips around the loop: a register is IN FLIGHT from the asm load that names it as destination to the asm wait tagged
# "LANDED <regs>"; no instruction outside the asm blocks may read or write it in between
inflight, bad, inasm = set(), [], False
for trip in range(2):          # trip 0: prologue + first pass through the loop; trip 1: the loop again (loop-carried state)
    for i in (range(0, end + 1) if trip == 0 else range(hdr, end + 1)):
        raw = lines[i]
        if "#ASMSTART" in raw: inasm = True; continue
        if "#ASMEND" in raw: inasm = False; continue
        code = raw.split(";")[0].strip()
        if inasm:
            if "LANDED" in raw:
                for c, r in regs(raw.split("LANDED")[1]): inflight.discard((c, r))
            elif code.startswith("buffer_load"):
                for c, r in regs(code.split(None, 1)[1].split(",")[0]): inflight.add((c, r))
            continue
        if not code or code.startswith("."): continue
        ops = code.split(None, 1)
        if len(ops) < 2: continue
        touched = [x for x in regs(ops[1]) if x in inflight]
        if touched: bad.append((trip, i, code))
print(f"main loop: lines {hdr}..{end}; compiler instructions touching a register with a load in flight: {len(bad)}")
for b in bad[:20]: print(" ", b)
sys.exit(1 if bad else 0)
