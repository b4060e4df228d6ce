#!/usr/bin/env python3
"""Where the time of one full-K IMAGE launch (gemm_fullk64.hip) goes: per-wave wall_clock64 stamps (100 MHz) at entry / requests out (helper: operands staged) /
first fragment computed / loop done / slices met / epilogue stores issued.  Tuning build with MI355_EXTRA_CFLAGS=-DMI355_FULLK_STAMPS only.
usage: fullk64_stamps.py [--shapes "K,nh,nkv,M;..."]   default: the QKV launch of Qwen2-7B (3584,28,4,64), its tp 4 shard (3584,7,1,64), Llama-3-70B tp 8 (8192,8,1,32)"""
import argparse, ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, kvcache, model, ops

ap = argparse.ArgumentParser(); ap.add_argument("--shapes", default="3584,28,4,64;3584,7,1,64;8192,8,1,32"); ap.add_argument("--set", default=""); a = ap.parse_args()
dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
lib = _C.lib()
for kv_ in [t for t in a.set.split(",") if t]:
    k_, v_ = kv_.split("="); lib.mi355_debug_set(int(k_), int(v_))
if not hasattr(lib, "mi355_debug_fullk_stamps"):
    sys.exit("the tuning library was built without the stamps: touch rtp_llm_amd/csrc/gemm_fullk.hip rtp_llm_amd/csrc/gemm_fullk64.hip && "
             "MI355_EXTRA_CFLAGS=-DMI355_FULLK_STAMPS python -m rtp_llm_amd.build --tuning")
lib.mi355_debug_fullk_stamps.argtypes = [C.c_void_p]
NB = 4096
st = torch.zeros(NB * 16 * 6, dtype=torch.int64, device=dev)
lib.mi355_debug_fullk_stamps(st.data_ptr())
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
COPIES = 5
hd, page, mbk, nblk = 128, 16, 160, 16384

def report(name, fn):
    for i in range(COPIES - 1):
        fn(i)
    flush.zero_(); torch.cuda.synchronize(); st.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(COPIES - 1); e1.record(); torch.cuda.synchronize()
    s = st.view(NB, 16, 6).cpu().double() * 0.01
    live = s[..., 0] > 0
    t0 = s[..., 0][live].min()
    def col(i, sub):
        v = s[..., i][sub & (s[..., i] > 0)] - t0
        return f"{v.mean():6.2f} (min {v.min():5.2f} max {v.max():5.2f})" if v.numel() else "   -"
    nw = int(live[live.any(1)][0].sum())
    helper = torch.zeros_like(live); helper[:, nw - 1] = live[:, nw - 1]
    kw = live & ~helper
    print(f"{name}: {int(live.any(1).sum())} blocks x {nw} waves (last = helper), events {e0.elapsed_time(e1) * 1e3:.1f} us")
    for i, lab in enumerate(["entry", "requests out", "first fragment", "loop done", "slices met", "epilogue out"]):
        print(f"    {lab:18s} K waves {col(i, kw)}   helper {col(i, helper)}")
    print(f"    last stamp at {(s[live.any(1)].max() - t0):.2f} us", flush=True)

for spec in a.shapes.split(";"):
    K, nh, nkv, M = (int(v) for v in spec.split(","))
    N = (nh + 2 * nkv) * hd
    wq = [model.synth_linear(K, N, "w4", dev, gen, zeros="centered").pack() for _ in range(COPIES)]
    cfg = model.ModelConfig("stamp", 1, K, nh, nkv, hd, 1024, 1024, max_pos=4096)
    cs = model.rope_table(cfg, dev)
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, dev)
    x = ops.act_image_pack((torch.randn(M, K, device=dev, generator=gen) * 0.5).half())
    pos = torch.full((M,), 1000, dtype=torch.int32, device=dev)
    bt = torch.arange(M * mbk, dtype=torch.int32, device=dev).reshape(M, mbk)
    print(f"==== K = {K}, heads {nh} q + {nkv} kv (N = {N}), M = {M}")
    report("qkv image launch (rope + kv write)", lambda i: ops.qkv_rope_kv_write_img(x, wq[i], None, cs, pos, bt, kv, sc, nh, nkv, hd, page))
    wo = [model.synth_linear(K, 8192 if K == 8192 else 3584, "w4", dev, gen, zeros="centered").pack() for _ in range(COPIES)]
    res = torch.randn(M, wo[0].N, device=dev, generator=gen).half()
    report("o-like image launch (residual)", lambda i: ops.linear_residual_img(x, wo[i], res))
