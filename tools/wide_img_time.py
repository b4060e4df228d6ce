#!/usr/bin/env python3
"""gate_up at 17-64 rows on the wide GEMM: row-major activations vs an activation image vs image + deferred RMSNorm (gemm_wide.hip),
Qwen2-7B shape, weights rotating through HBM-resident copies, graph replay.  usage: wide_img_time.py [--ms 64,32] [--tuning [--dbg 0,16,2,3]]
--dbg (tuning build, <= 32 rows): 0 = shipped (two chunks of weights requested ahead), 16 = one chunk ahead (rounds 2-4), 2 = the instruction
stream without weight traffic, 3 = without any main-loop traffic."""
import argparse, os, sys
ap = argparse.ArgumentParser(); ap.add_argument("--ms", default="64,32"); ap.add_argument("--iters", type=int, default=30); ap.add_argument("--tuning", action="store_true"); ap.add_argument("--dbg", default="")
ap.add_argument("--resident", action="store_true", help="also ONE weight copy (72 MB: stays in the Infinity Cache) and a HALF-resident mix")
a = ap.parse_args()
if a.tuning:
    os.environ["MI355_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, model, ops
dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
cfg = model.QWEN2_7B; H, I = cfg.hidden, cfg.inter
wg = [model.synth_linear(H, 2 * I, "w4", dev, gen, zeros="centered").pack(gate_up=True) for _ in range(8)]

def timed(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for r in range(a.iters): fn(r)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.iters

for M in [int(m) for m in a.ms.split(",")]:
    x = (torch.randn(M, H, device=dev, generator=gen) * 0.5).half()
    xi = ops.act_image_pack(x)
    ssq = torch.rand(M, (H // 16 + 3) & ~3, device=dev, generator=gen)
    t = [timed(lambda i: ops.linear(x, wg[i % 8], None, _C.EPI_SILU_MUL), 8),
         timed(lambda i: ops.linear_deferred_norm_img(xi, None, wg[i % 8], None, _C.EPI_SILU_MUL), 8),
         timed(lambda i: ops.linear_deferred_norm_img(xi, (ssq, 1e-6, 1), wg[i % 8], None, _C.EPI_SILU_MUL), 8)]
    if a.tuning and a.dbg:
        import ctypes as C
        lib = _C.lib(); lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]; lib.mi355_debug_set.restype = None
        for v in [int(v) for v in a.dbg.split(",")]:
            lib.mi355_debug_set(0, v)
            tv = timed(lambda i: ops.linear_deferred_norm_img(xi, (ssq, 1e-6, 1), wg[i % 8], None, _C.EPI_SILU_MUL), 8)
            print(f"M={M:3d}  image + deferred norm, switch {v:2d}: {tv:6.2f} us", flush=True)
        lib.mi355_debug_set(0, 0)
    if a.resident:
        tr = timed(lambda i: ops.linear_deferred_norm_img(xi, (ssq, 1e-6, 1), wg[0], None, _C.EPI_SILU_MUL), 4)
        print(f"M={M:3d}  image + deferred norm, ONE weight copy (cache-resident): {tr:6.2f} us", flush=True)
    print(f"M={M:3d}  gate_up + SiLU: row-major {t[0]:6.2f}   image {t[1]:6.2f}   image + deferred norm {t[2]:6.2f} us (graph replay, gaps included)", flush=True)
