#!/usr/bin/env python3
"""Where the time of one staged split-K GEMM launch (gemm_wq_kernel) goes: wall_clock64 stamps (100 MHz) of wave 0 of every block
at entry / after the prologue barrier / after the main loop / after the slab stores were issued / after they drained.
usage: wq_stamps.py [--shapes qkv,o,down] [--m 64]   (tuning build; switch 7 = 2 arms the stamps)"""
import argparse, ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, model, ops  # noqa: E402

SH = {"qkv": (3584, 4608), "o": (3584, 3584), "down": (18944, 3584)}
ap = argparse.ArgumentParser(); ap.add_argument("--shapes", default="qkv,o,down"); ap.add_argument("--m", type=int, default=64)
a = ap.parse_args()
lib = _C.lib(); dev = "cuda:0"
lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]; lib.mi355_debug_ptr.argtypes = [C.c_void_p]
gen = torch.Generator(device=dev).manual_seed(0)
st = torch.zeros(4096 * 5, dtype=torch.int64, device=dev)
lib.mi355_debug_ptr(st.data_ptr()); lib.mi355_debug_set(7, 2)
for name in a.shapes.split(","):
    K, N = SH[name]; M = a.m
    ws = [model.synth_linear(K, N, "w4", dev, gen).pack() for _ in range(3)]
    x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
    slabs = torch.empty(16 * M * ws[0].N_pad, dtype=torch.float32, device=dev)
    big = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
    for i in range(3):
        big.fill_(i)            # flush the Infinity Cache between launches
        torch.cuda.synchronize()
        st.zero_()
        n = lib.mi355_linear_partial(x.data_ptr(), M, C.byref(ops.weight_struct(ws[i])), slabs.data_ptr(), 16, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    s = st.view(-1, 5).cpu().double() * 0.01
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    d = lambda i, j: (s[:, j] - s[:, i])
    print(f"{name} M={M}: {n} slabs, {s.shape[0]} blocks; block starts spread over {s[:, 0].max() - t0:.2f} us; last block exits at {s[:, 4].max() - t0:.2f} us")
    for lbl, i, j in (("prologue (entry -> first chunk staged)", 0, 1), ("main loop", 1, 2), ("merge + slab stores issued", 2, 3), ("store drain (vmcnt 0)", 3, 4)):
        v = d(i, j)
        print(f"   {lbl:42s} mean {v.mean():6.2f}  min {v.min():6.2f}  max {v.max():6.2f} us")
