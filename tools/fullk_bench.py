#!/usr/bin/env python3
"""Launch the fused full-K ops (and the launches they replace) a few times at the Qwen2-7B shapes; run under
`rocprofv3 --kernel-trace` (tools/probe/ktrace.sh) to read kernel durations.  usage: fullk_bench.py [--ms 1,16,64]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, kvcache, model, ops

ap = argparse.ArgumentParser(); ap.add_argument("--ms", default="1,16,64"); ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--old", type=int, default=1); a = ap.parse_args()
dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
cfg = model.QWEN2_7B
nh, nkv, hd, H, I = cfg.nh, cfg.nkv, cfg.hd, cfg.hidden, cfg.inter
wq = model.synth_linear(H, (nh + 2 * nkv) * hd, "w4", dev, gen, zeros="centered").pack()
wo = model.synth_linear(H, H, "w4", dev, gen, zeros="centered").pack()
wd = model.synth_linear(I, H, "w4", dev, gen, zeros="centered").pack()
page, mb, nblk = 16, 64, 4096
cs = model.rope_table(cfg, dev)
kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, dev)
for M in [int(m) for m in a.ms.split(",")]:
    x = (torch.randn(M, H, device=dev, generator=gen) * 0.5).half()
    act = (torch.randn(M, I, device=dev, generator=gen) * 0.5).half()
    res = torch.randn(M, H, device=dev, generator=gen).half()
    pos = torch.full((M,), 1000, dtype=torch.int32, device=dev)
    bt = torch.arange(M * mb, dtype=torch.int32, device=dev).reshape(M, mb)
    for _ in range(a.iters):
        ops.qkv_rope_kv_write(x, wq, None, cs, pos, bt, kv, sc, nh, nkv, hd, page)
        ops.linear_residual(x, wo, res)
        ops.linear_residual(act, wd, res)
        if a.old:
            y = ops.linear(x, wq); ops.rope_kv_write_rows(y, None, cs, pos, bt, kv, sc, nh, nkv, hd, page, 1)
            ops.linear(x, wo); ops.linear(act, wd)
    torch.cuda.synchronize()
    print("done M", M, flush=True)
