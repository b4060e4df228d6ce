#!/usr/bin/env python3
"""A/B of gemm_wide debug variants (tuning build): output of variant V must be bit-identical to variant 0; prints timings.
usage: wide_var_check.py [--vars 0,5,6,7] [--m 64] [--shape gate_up]"""
import argparse, ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, model, ops  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--vars", default="0,5,6,7"); ap.add_argument("--m", type=int, default=64)
ap.add_argument("--shape", default="gate_up"); ap.add_argument("--iters", type=int, default=60)
a = ap.parse_args()
SH = {"gate_up": (3584, 37888), "down": (18944, 3584), "gu70": (8192, 57344)}
K, N = SH[a.shape]; M = a.m; dev = "cuda:0"
lib = _C.lib(); lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]
gen = torch.Generator(device=dev).manual_seed(0)
base = model.synth_linear(K, N, "w4", dev, gen).pack(gate_up=a.shape != "down")
ncopy = max(2, int(600e6 // base.nbytes) + 1)
copies = [base] + [type(base)(base.qweight.clone(), base.meta.clone(), base.wbits, base.K, base.N, base.K_pad, base.N_pad, base.group_size) for _ in range(ncopy - 1)]
x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
epi = _C.EPI_SILU_MUL if a.shape != "down" else 0
ref = None
for rnd in range(2):
    for v in [int(t) for t in a.vars.split(",")]:
        lib.mi355_debug_set(0, v)
        y = ops.linear(x, base, None, epi); torch.cuda.synchronize()
        if v == 0 and ref is None: ref = y.clone()
        same = torch.equal(y, ref)
        out = torch.empty_like(y)
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(8): ops.linear(x, copies[i % ncopy], None, epi, out=out)
        st.record()
        for i in range(a.iters): ops.linear(x, copies[i % ncopy], None, epi, out=out)
        en.record(); torch.cuda.synchronize()
        print(f"{a.shape} M={M} var={v}: {st.elapsed_time(en) / a.iters * 1e3:7.2f} us   bit-identical to var 0: {same}", flush=True)
lib.mi355_debug_set(0, 0)
