#!/usr/bin/env python3
"""down_proj at 17-64 rows: the K-quarter launch on an activation image (gemm_splitk64.hip) + its fold vs the staged split-K kernel
(gemm.hip) + its fold, Qwen2-7B shape, weights rotating through HBM-resident copies, graph replay.  usage: splitk64_time.py [--ms 64,32]"""
import argparse, ctypes as C, os, sys
ap = argparse.ArgumentParser(); ap.add_argument("--ms", default="64,48,32,17"); ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--set", default="", help="k=v,... -> mi355_debug_set (tuning build): 7=1 no activation traffic, 7=2 no weight traffic, 7=3 neither")
ap.add_argument("--resident", action="store_true", help="also ONE weight copy (36 MB: stays in the Infinity Cache): what a perfect weight prefetch would buy")
a = ap.parse_args()
if a.set: os.environ["MI355_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, model, ops
dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
cfg = model.QWEN2_7B; H, I = cfg.hidden, cfg.inter
wd = [model.synth_linear(I, H, "w4", dev, gen, zeros="centered").pack() for _ in range(10)]
lib = _C.lib()
for kv_ in [x for x in a.set.split(",") if x]:
    lib.mi355_debug_set(int(kv_.split("=")[0]), int(kv_.split("=")[1]))

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
    x = (torch.randn(M, I, device=dev, generator=gen) * 0.5).half()
    xi = ops.act_image_pack(x)
    res = torch.randn(M, H, device=dev, generator=gen).half(); gamma = torch.ones(H, dtype=torch.float16, device=dev)
    slabs = torch.empty(16, M, wd[0].N_pad, dtype=torch.float32, device=dev)
    y = torch.empty(M, H, dtype=torch.float16, device=dev); r2 = torch.empty_like(res)
    st = lambda: torch.cuda.current_stream().cuda_stream
    ns = [0, 0]
    def new(i, fold):
        w = ops.weight_struct(wd[i % 10])
        ns[0] = lib.mi355_linear_partial_img(xi.data.data_ptr(), M, C.byref(w), slabs.data_ptr(), 16, st())
        if fold: lib.mi355_add_rmsnorm(None, slabs.data_ptr(), ns[0], wd[0].N_pad, None, res.data_ptr(), r2.data_ptr(), gamma.data_ptr(), 1e-6, M, H, y.data_ptr(), st())
    def old(i, fold):
        w = ops.weight_struct(wd[i % 10])
        ns[1] = lib.mi355_linear_partial(x.data_ptr(), M, C.byref(w), slabs.data_ptr(), 16, st())
        if fold: lib.mi355_add_rmsnorm(None, slabs.data_ptr(), ns[1], wd[0].N_pad, None, res.data_ptr(), r2.data_ptr(), gamma.data_ptr(), 1e-6, M, H, y.data_ptr(), st())
    t = [timed(lambda i: new(i, False), 10), timed(lambda i: new(i, True), 10), timed(lambda i: old(i, False), 10), timed(lambda i: old(i, True), 10)]
    if a.resident:
        tr = timed(lambda i: new(0, False), 4)
        print(f"M={M:3d}  down: image K-quarters, ONE weight copy (cache-resident) {tr:6.2f} us", flush=True)
    print(f"M={M:3d}  down: image K-quarters {t[0]:6.2f} ({ns[0]} slabs), + fold {t[1]:6.2f}   |   staged {t[2]:6.2f} ({ns[1]} slabs), + fold {t[3]:6.2f} us (graph replay, gaps included)", flush=True)
