#!/usr/bin/env python3
"""Where the time of one full-K launch goes at a few rows: per-wave wall_clock64 stamps (100 MHz) at entry / requests out
(+ norm meet) / first half chunk computed (= first weights landed) / loop done / slices met / epilogue stores issued.
Tuning build with MI355_EXTRA_CFLAGS=-DMI355_FULLK_STAMPS only.  usage: fullk_stamps.py [--ms 1,8]"""
import argparse, ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, kvcache, model, ops

ap = argparse.ArgumentParser(); ap.add_argument("--ms", default="1,8"); ap.add_argument("--set", default=""); a = ap.parse_args()
dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
cfg = model.QWEN2_7B
nh, nkv, hd, H, I = cfg.nh, cfg.nkv, cfg.hd, cfg.hidden, cfg.inter
COPIES = 5   # rotate weights so that every launch streams from HBM
mk = lambda K, N, **kw: [model.synth_linear(K, N, "w4", dev, gen, zeros="centered").pack(**kw) for _ in range(COPIES)]
wq, wo, wd, wg = mk(H, (nh + 2 * nkv) * hd), mk(H, H), mk(I, H), mk(H, 2 * I, gate_up=True)
page, mbk, nblk = 16, 64, 4096
cs = model.rope_table(cfg, dev)
kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, dev)
gamma = torch.ones(H, dtype=torch.float16, device=dev)
lib = _C.lib()
for kv_ in [t for t in a.set.split(",") if t]:
    k_, v_ = kv_.split("="); lib.mi355_debug_set(int(k_), int(v_))
if not hasattr(lib, "mi355_debug_fullk_stamps"):
    sys.exit("the tuning library was built without the stamps: touch rtp_llm_amd/csrc/gemm_fullk.hip && "
             "MI355_EXTRA_CFLAGS=-DMI355_FULLK_STAMPS python -m rtp_llm_amd.build --tuning  (the stamp stores change the schedule: rebuild without the flag afterwards)")
lib.mi355_debug_fullk_stamps.argtypes = [C.c_void_p]
NB = 4096
st = torch.zeros(NB * 16 * 6, dtype=torch.int64, device=dev)
lib.mi355_debug_fullk_stamps(st.data_ptr())
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

def report(name, fn):
    for i in range(COPIES - 1):
        fn(i)
    flush.zero_(); torch.cuda.synchronize(); st.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(COPIES - 1); e1.record(); torch.cuda.synchronize()
    s = st.view(NB, 16, 6).cpu().double() * 0.01
    live = s[..., 0] > 0
    t0 = s[..., 0][live].min()
    def col(i, sub=live):
        v = s[..., i][sub & (s[..., i] > 0)] - t0
        return f"{v.mean():6.2f} (min {v.min():5.2f} max {v.max():5.2f})" if v.numel() else "   -"
    print(f"{name}: {int(live.any(1).sum())} blocks x {int(live[live.any(1)][0].sum())} waves, events {e0.elapsed_time(e1) * 1e3:.1f} us")
    for i, lab in enumerate(["entry", "requests out", "first half chunk", "loop done", "slices met", "epilogue out"]):
        print(f"    {lab:18s} {col(i)}")
    print(f"    last stamp at {(s[live.any(1)].max() - t0):.2f} us", flush=True)

for M in [int(m) for m in a.ms.split(",")]:
    print(f"==== M = {M}")
    x = (torch.randn(M, H, device=dev, generator=gen) * 0.5).half()
    act = (torch.randn(M, I, device=dev, generator=gen) * 0.5).half()
    res = torch.randn(M, H, device=dev, generator=gen).half()
    ssq = torch.zeros(M, H // 16, dtype=torch.float32, device=dev)
    pos = torch.full((M,), 1000, dtype=torch.int32, device=dev)
    bt = torch.arange(M * mbk, dtype=torch.int32, device=dev).reshape(M, mbk)
    ops.linear_residual(x, wo[0], res, tile_sumsq=ssq)
    norm = (ssq, gamma, 1e-6)
    if M > 16:   # gemm_fullk64.hip (or, with --set 5=2, the generic kernel at these heights): no fused norm, QKV and O only
        report("qkv (rope + kv write)", lambda i: ops.qkv_rope_kv_write(res, wq[i], None, cs, pos, bt, kv, sc, nh, nkv, hd, page))
        report("o (residual)", lambda i: ops.linear_residual(x, wo[i], res))
        continue
    report("qkv (norm + rope + kv write)", lambda i: ops.qkv_rope_kv_write(res, wq[i], None, cs, pos, bt, kv, sc, nh, nkv, hd, page, norm=norm))
    report("o (residual)", lambda i: ops.linear_residual(x, wo[i], res))
    report("gate_up (norm + silu)", lambda i: ops.norm_linear(res, norm, wg[i], None, _C.EPI_SILU_MUL))
    report("down (residual)", lambda i: ops.linear_residual(act, wd[i], res))
