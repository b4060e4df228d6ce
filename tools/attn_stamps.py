#!/usr/bin/env python3
"""Where the time of one paged-attention launch goes: per-wave wall_clock64 stamps (100 MHz) at entry / first requests out / first
group computed / loop done / waves met / stores issued.  Tuning build.  usage: attn_stamps.py [--batch 64 --ctx 1024 --int8 --copies N]"""
import argparse, ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, kvcache, ops
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--ctx", type=int, default=1024)
ap.add_argument("--int8", action="store_true"); ap.add_argument("--page", type=int, default=16); ap.add_argument("--copies", type=int, default=4)
a = ap.parse_args()
lib = _C.lib(); lib.mi355_debug_attn_stamps.argtypes = [C.c_void_p]
dev = "cuda:0"; nh, nkv, hd = 28, 4, 128
B, ctx, page = a.batch, a.ctx, a.page
mb = (ctx + page - 1) // page; nblk = B * mb
g = torch.Generator(device=dev).manual_seed(0)
caches = []
for _ in range(a.copies):
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, a.int8, dev)
    if a.int8:
        kv.copy_(torch.randint(-127, 128, kv.shape, device=dev, generator=g, dtype=torch.int8)); sc.uniform_(0.005, 0.02)
    else:
        kv.copy_(torch.randn(kv.shape, device=dev, generator=g, dtype=torch.float16))
    caches.append((kv, sc))
bt = torch.randperm(nblk, generator=torch.Generator().manual_seed(1)).reshape(B, mb).to(torch.int32).to(dev)
sl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
q = torch.randn(B, nh, hd, device=dev, generator=g, dtype=torch.float16)
NB = B * nkv * 8
st = torch.zeros(NB * 4 * 6, dtype=torch.int64, device=dev)
for i in range(a.copies):
    ops.paged_decode_attention(q, caches[i][0], caches[i][1], bt, sl, nkv, page, ctx)
torch.cuda.synchronize()
for rep in range(2):
    lib.mi355_debug_attn_stamps(st.data_ptr()); st.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kv, sc = caches[rep % a.copies]
    e0.record(); ops.paged_decode_attention(q, kv, sc, bt, sl, nkv, page, ctx); e1.record(); torch.cuda.synchronize()
    lib.mi355_debug_attn_stamps(None)
    s = st.view(-1, 4, 6).cpu().double() * 0.01
    live = s[..., 0] > 0
    t0 = s[..., 0][live].min()
    print(f"attention B={B} ctx={ctx} int8={a.int8}: {int(live.any(1).sum())} blocks, events {e0.elapsed_time(e1) * 1e3:.1f} us" + (" (cache copy last used %d launches ago)" % a.copies))
    for i, lab in enumerate(["entry", "first requests out", "first group computed", "loop done", "waves met", "stores issued"]):
        v = s[..., i][live & (s[..., i] > 0)] - t0
        print(f"    {lab:22s} mean {v.mean():6.2f}  min {v.min():6.2f}  max {v.max():6.2f}" if v.numel() else f"    {lab:22s} -")
