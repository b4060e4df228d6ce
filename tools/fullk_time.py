#!/usr/bin/env python3
"""Time the fused full-K launches of the few-row decode layer (QKV + norm + RoPE + KV write, O + residual, gate_up + norm + SiLU,
down + residual) at the Qwen2-7B shapes, weights rotating through HBM-resident copies.  --product: the product library
(default: the tuning build, so that a previous source state can be kept there for a same-box A/B).
usage: fullk_time.py [--product] [--ms 1,4,8,16]"""
import argparse, os, sys
ap = argparse.ArgumentParser(); ap.add_argument("--ms", default="1,4,8,16"); ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--product", action="store_true"); ap.add_argument("--set", default="", help="k=v,... -> mi355_debug_set (tuning build)")
ap.add_argument("--sweep", default="", help="k=v1,v2,...: repeat the measurement for every value of debug key k (tuning build)")
ap.add_argument("--resident", action="store_true", help="also ONE weight copy per linear (cache-resident): what a perfect weight prefetch would buy")
a = ap.parse_args()
if not a.product:
    os.environ["MI355_TUNING_LIB"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, kvcache, model, ops

dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
cfg = model.QWEN2_7B
nh, nkv, hd, H, I = cfg.nh, cfg.nkv, cfg.hd, cfg.hidden, cfg.inter
COPIES = 8
mk = lambda K, N, n=COPIES, **kw: [model.synth_linear(K, N, "w4", dev, gen, zeros="centered").pack(**kw) for _ in range(n)]
wq, wo, wd, wg = mk(H, (nh + 2 * nkv) * hd, 40), mk(H, H, 48), mk(I, H), mk(H, 2 * I, gate_up=True)     # > 256 MB per set: past the Infinity Cache
page, mbk, nblk = 16, 64, 4096
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

for kv_ in [t for t in a.set.split(",") if t]:
    k_, v_ = kv_.split("="); _C.lib().mi355_debug_set(int(k_), int(v_))
sweep_k, sweep_v = (int(a.sweep.split("=")[0]), [int(v) for v in a.sweep.split("=")[1].split(",")]) if a.sweep else (None, [None])
print("lib:", "product" if a.product else "tuning", a.set, a.sweep)
for M in [int(m) for m in a.ms.split(",")]:
    x = (torch.randn(M, H, device=dev, generator=gen) * 0.5).half()
    act = (torch.randn(M, I, device=dev, generator=gen) * 0.5).half()
    res = torch.randn(M, H, device=dev, generator=gen).half()
    ssq = torch.zeros(M, H // 16, dtype=torch.float32, device=dev)
    pos = torch.full((M,), 1000, dtype=torch.int32, device=dev)
    bt = torch.arange(M * mbk, dtype=torch.int32, device=dev).reshape(M, mbk)
    ops.linear_residual(x, wo[0], res, tile_sumsq=ssq)
    norm = (ssq, gamma, 1e-6)
    out = torch.empty_like(res)
    for sv in sweep_v:
        if sv is not None:
            _C.lib().mi355_debug_set(sweep_k, sv)
        t = [timed(lambda i: ops.qkv_rope_kv_write(res, wq[i % len(wq)], None, cs, pos, bt, kv, sc, nh, nkv, hd, page, norm=norm), len(wq)),
             timed(lambda i: ops.linear_residual(x, wo[i % len(wo)], res, out=out), len(wo)),
             timed(lambda i: ops.norm_linear(res, norm, wg[i % len(wg)], None, _C.EPI_SILU_MUL), len(wg)),
             timed(lambda i: ops.linear_residual(act, wd[i % len(wd)], res, out=out), len(wd))]
        if a.resident:
            tr = [timed(lambda i: ops.qkv_rope_kv_write(res, wq[0], None, cs, pos, bt, kv, sc, nh, nkv, hd, page, norm=norm), 4),
                  timed(lambda i: ops.linear_residual(x, wo[0], res, out=out), 4),
                  timed(lambda i: ops.norm_linear(res, norm, wg[0], None, _C.EPI_SILU_MUL), 4),
                  timed(lambda i: ops.linear_residual(act, wd[0], res, out=out), 4)]
            print(f"M={M:3d} [one weight copy: cache-resident]  qkv {tr[0]:6.2f}  o {tr[1]:6.2f}  gate_up {tr[2]:6.2f}  down {tr[3]:6.2f}  sum {sum(tr):6.2f} us", flush=True)
        tag = "" if sv is None else f" [{sweep_k}={sv}]"
        print(f"M={M:3d}{tag}  qkv {t[0]:6.2f}  o {t[1]:6.2f}  gate_up {t[2]:6.2f}  down {t[3]:6.2f}  sum {sum(t):6.2f} us (graph replay, launch gaps included)", flush=True)
