#!/usr/bin/env python3
"""Time the 17-64-row full-K launches (gemm_fullk64.hip: QKV + bias + RoPE + KV write, O + residual) at the Qwen2-7B shapes,
weights rotating through HBM-resident copies, graph-replayed.  Tuning library by default, so that the experiment switches work:
  6=1 / 6=2  one / three k-steps of activations in flight     7=1  no activation traffic   7=2  no weight traffic   7=3  neither
  (same instruction stream).  The last line per M: the row-major generic full-K kernel and the composed launches.
usage: fullk64_time.py [--product] [--ms 64,32,17] [--variants "5=2;6=1;6=2;7=1;7=2;7=3"]"""
import argparse, os, sys
ap = argparse.ArgumentParser(); ap.add_argument("--ms", default="64,32,17"); ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--product", action="store_true"); ap.add_argument("--variants", default="")
ap.add_argument("--resident", action="store_true", help="also time with ONE weight copy (stays in the Infinity Cache / L2): what a perfect weight prefetch would buy")
a = ap.parse_args()
if not a.product:
    os.environ["MI355_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, kvcache, model, ops

dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
cfg = model.QWEN2_7B
nh, nkv, hd, H, I = cfg.nh, cfg.nkv, cfg.hd, cfg.hidden, cfg.inter
mk = lambda K, N, n, **kw: [model.synth_linear(K, N, "w4", dev, gen, zeros="centered").pack(**kw) for _ in range(n)]
wq, wo = mk(H, (nh + 2 * nkv) * hd, 40), mk(H, H, 48)     # > 256 MB per set: past the Infinity Cache
page, mbk, nblk = 16, 64, 8192
cs = model.rope_table(cfg, dev)
kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, dev)
gamma = torch.ones(H, dtype=torch.float16, device=dev)

def timed(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for r in range(a.iters):
            fn(r)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.iters

def setv(spec, on):
    for kv_ in [t for t in spec.split(",") if t]:
        k_, v_ = kv_.split("="); _C.lib().mi355_debug_set(int(k_), int(v_) if on else 0)

print("lib:", "product" if a.product else "tuning")
for M in [int(m) for m in a.ms.split(",")]:
    x = (torch.randn(M, H, device=dev, generator=gen) * 0.5).half()
    res = torch.randn(M, H, device=dev, generator=gen).half()
    pos = torch.full((M,), 1000, dtype=torch.int32, device=dev)
    bt = torch.arange(M * mbk, dtype=torch.int32, device=dev).reshape(M, mbk)
    out = torch.empty_like(res)
    xi = ops.act_image_pack(x)
    ssq = torch.zeros(M, H // 16, dtype=torch.float32, device=dev)
    for var in [""] + ([] if a.product else [v for v in a.variants.split(";") if v]):
        setv(var, True)
        t = [timed(lambda i: ops.qkv_rope_kv_write_img(xi, wq[i % len(wq)], None, cs, pos, bt, kv, sc, nh, nkv, hd, page), len(wq)),
             timed(lambda i: ops.linear_residual_img(xi, wo[i % len(wo)], res, out=out, tile_sumsq=ssq), len(wo))]
        setv(var, False)
        print(f"M={M:3d} [{var or 'default':8s}]  qkv+rope+kv {t[0]:6.2f}  o+residual {t[1]:6.2f} us (graph replay, launch gaps included)", flush=True)
        if a.resident and not var:
            t = [timed(lambda i: ops.qkv_rope_kv_write_img(xi, wq[0], None, cs, pos, bt, kv, sc, nh, nkv, hd, page), 4),
                 timed(lambda i: ops.linear_residual_img(xi, wo[0], res, out=out, tile_sumsq=ssq), 4)]
            print(f"M={M:3d} [resident]  qkv+rope+kv {t[0]:6.2f}  o+residual {t[1]:6.2f} us (one weight copy: cache-resident)", flush=True)
    t = [timed(lambda i: ops.qkv_rope_kv_write(x, wq[i % len(wq)], None, cs, pos, bt, kv, sc, nh, nkv, hd, page), len(wq)),
         timed(lambda i: ops.linear_residual(x, wo[i % len(wo)], res, out=out, tile_sumsq=ssq), len(wo))]
    print(f"M={M:3d} [row-major]  qkv+rope+kv {t[0]:6.2f}  o+residual {t[1]:6.2f} us (gemm_fullk.hip, fragments gathered from the row-major tensor)", flush=True)
    # the composed launches they replace
    y = torch.empty(M, (nh + 2 * nkv) * hd, dtype=torch.float16, device=dev)
    t = [timed(lambda i: ops.rope_kv_write_rows(ops.linear(x, wq[i % len(wq)], None), None, cs, pos, bt, kv, sc, nh, nkv, hd, page, 1), len(wq)),
         timed(lambda i: ops.add_rmsnorm(ops.linear(x, wo[i % len(wo)], None), res, gamma, 1e-6), len(wo))]
    print(f"M={M:3d} [composed]  linear + rope_kv_write {t[0]:6.2f}  linear + add_rmsnorm {t[1]:6.2f} us (the stand-alone API: fp16 tensor between the two launches, not the engine's slab fold)", flush=True)
