#!/usr/bin/env python3
"""Micro-benchmark of paged decode attention through the C-ABI; several KV copies rotated (HBM-resident).
usage: attn_bench.py [--batch 64 --ctx 1024 --int8 --ps N]"""
import argparse, ctypes as C, os, sys
import torch
if "--product" not in sys.argv:
    os.environ["MI355_TUNING_LIB"] = "1"   # experiment switches live in the tuning build only (python -m rtp_llm_amd.build --tuning)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, kvcache, ops  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64); ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--int8", action="store_true"); ap.add_argument("--ps", type=int, default=0)
    ap.add_argument("--page", type=int, default=16); ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--seq-bt", type=int, default=0, help="1: identity block table (contiguous pages) instead of a random permutation")
    ap.add_argument("--product", action="store_true", help="time the product library instead of the tuning build")
    ap.add_argument("--tune", default="", help="idx=val,... forwarded to mi355_debug_set")
    ap.add_argument("--copies", type=int, default=0, help="KV copies rotated (default: enough to stay HBM-resident; 1 = one copy that may live in the Infinity Cache)")
    a = ap.parse_args()
    lib = _C.lib()
    if not a.product:
        lib.mi355_debug_set_attn.argtypes = [C.c_int]; lib.mi355_debug_set_attn(a.ps)
        lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]
    for kv_ in filter(None, a.tune.split(",")):
        lib.mi355_debug_set(int(kv_.split("=")[0]), int(kv_.split("=")[1]))
    dev = "cuda:0"; nh, nkv, hd = 28, 4, 128
    B, ctx, page = a.batch, a.ctx, a.page
    mb = (ctx + page - 1) // page
    nblk = B * mb
    g = torch.Generator(device=dev).manual_seed(0)
    bytes_kv = B * ctx * 2 * nkv * hd * (1 if a.int8 else 2)
    ncopy = a.copies or max(2, int(700e6 // bytes_kv) + 1)
    caches = []
    for _ in range(ncopy):
        kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, a.int8, dev)
        if a.int8:
            kv.copy_(torch.randint(-127, 128, kv.shape, device=dev, generator=g, dtype=torch.int8)); sc.uniform_(0.005, 0.02)
        else:
            kv.copy_(torch.randn(kv.shape, device=dev, generator=g, dtype=torch.float16))
        caches.append((kv, sc))
    bt = (torch.arange(nblk) if a.seq_bt else torch.randperm(nblk, generator=torch.Generator().manual_seed(1))).reshape(B, mb).to(torch.int32).to(dev)
    sl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    q = torch.randn(B, nh, hd, device=dev, generator=g, dtype=torch.float16)
    for kv, sc in caches[:2]:
        ops.paged_decode_attention(q, kv, sc, bt, sl, nkv, page, ctx)
    torch.cuda.synchronize()
    if a.tune:   # the switched kernel against the default one on the same cache (ragged contexts included)
        slr = torch.randint(1, ctx + 1, (B,), generator=torch.Generator().manual_seed(2), dtype=torch.int32).to(dev)
        for lens in (sl, slr):
            got = ops.paged_decode_attention(q, caches[0][0], caches[0][1], bt, lens, nkv, page, ctx).clone()
            for kv_ in filter(None, a.tune.split(",")):
                lib.mi355_debug_set(int(kv_.split("=")[0]), 0)
            ref = ops.paged_decode_attention(q, caches[0][0], caches[0][1], bt, lens, nkv, page, ctx).clone()
            for kv_ in filter(None, a.tune.split(",")):
                lib.mi355_debug_set(int(kv_.split("=")[0]), int(kv_.split("=")[1]))
            torch.cuda.synchronize()
            print(f"  tune [{a.tune}] vs default: max |diff| {float((got.float() - ref.float()).abs().max()):.3e}, bit-equal {bool(torch.equal(got, ref))}")
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(a.iters):
        kv, sc = caches[i % ncopy]
        ops.paged_decode_attention(q, kv, sc, bt, sl, nkv, page, ctx)
    en.record(); torch.cuda.synchronize()
    us = st.elapsed_time(en) / a.iters * 1e3
    print(f"attn B={B} ctx={ctx} int8={a.int8} ps={a.ps or 'auto'} page={page}: {us:8.2f} us  {bytes_kv / us / 1e3:8.1f} GB/s ({bytes_kv/1e6:.1f} MB, {ncopy} copies)", flush=True)

if __name__ == "__main__":
    main()
