#!/usr/bin/env python3
"""Split-K seam micro-benchmark: [linear_partial -> consumer] pairs as the engine issues them at <= 64 rows, captured into one
hipGraph (weights rotated so the Infinity Cache cannot hold them) and replayed: device time per pair incl. the kernel boundary.
  qkv  -> mi355_rope_kv_write_rows (slab fold + bias + RoPE + KV write)
  o    -> mi355_add_rmsnorm (slab fold + residual + RMSNorm)
  down -> mi355_add_rmsnorm
usage: seam_bench.py [--ms 64] [--shapes qkv,o,down] [--nsplits 0,4,7] [--cfgs 0,5,9] [--tune i=v,...]   (0 = planner's choice)"""
import argparse, ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, kvcache, model, ops  # noqa: E402

SH = {"qkv": (3584, 4608), "o": (3584, 3584), "down": (18944, 3584)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="64"); ap.add_argument("--shapes", default="qkv,o,down")
    ap.add_argument("--nsplits", default="0"); ap.add_argument("--cfgs", default="0"); ap.add_argument("--tune", default="")
    ap.add_argument("--pairs", type=int, default=24); ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    lib = _C.lib(); lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]
    for kv_ in filter(None, a.tune.split(",")):
        lib.mi355_debug_set(int(kv_.split("=")[0]), int(kv_.split("=")[1]))
    dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
    nh, nkv, hd, page = 28, 4, 128, 16
    for name in a.shapes.split(","):
        K, N = SH[name]
        base = model.synth_linear(K, N, "w4", dev, gen).pack()
        ncopy = max(2, int(600e6 // base.nbytes) + 1)
        copies = [base] + [type(base)(base.qweight.clone(), base.meta.clone(), base.wbits, base.K, base.N, base.K_pad, base.N_pad, base.group_size)
                           for _ in range(ncopy - 1)]
        structs = [ops.weight_struct(c) for c in copies]
        for M in [int(m) for m in a.ms.split(",")]:
            x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
            slabs = torch.empty(16 * M * base.N_pad, dtype=torch.float32, device=dev)
            resid = torch.randn(M, 3584, device=dev, generator=gen).half(); xn = torch.empty_like(resid)
            wn = torch.ones(3584, device=dev).half()
            kvb, _ = kvcache.alloc_layer_cache(M * 4, nkv, page, hd, False, dev)
            kvs = ops.kv_struct(kvb, None, page, nkv, hd)
            cs = torch.randn(2048, hd // 2, 2, device=dev)
            pos = torch.full((M,), 17, dtype=torch.int32, device=dev)
            bt = torch.arange(M * 4, dtype=torch.int32, device=dev).reshape(M, 4)
            qout = torch.empty(M, nh, hd, dtype=torch.float16, device=dev)
            bias = torch.zeros(N, device=dev).half()
            oob = torch.zeros(64, dtype=torch.int32, device=dev)
            for ns in [int(v) for v in a.nsplits.split(",")]:
                for cfg in [int(v) for v in a.cfgs.split(",")]:
                    lib.mi355_debug_set(1, ns); lib.mi355_debug_set(2, cfg)
                    st = torch.cuda.Stream()
                    def pair(i, s):
                        n = lib.mi355_linear_partial(x.data_ptr(), M, C.byref(structs[i % ncopy]), slabs.data_ptr(), 16, s)
                        assert n > 0, n
                        if name == "qkv":
                            rc = lib.mi355_rope_kv_write_rows(None, slabs.data_ptr(), n, base.N_pad, bias.data_ptr(), cs.data_ptr(), hd, 2048, pos.data_ptr(),
                                                              bt.data_ptr(), 4, M, 1, nh, C.byref(kvs), qout.data_ptr(), oob.data_ptr(), s)
                        else:
                            rc = lib.mi355_add_rmsnorm(None, slabs.data_ptr(), n, base.N_pad, None, resid.data_ptr(), resid.data_ptr(), wn.data_ptr(), 1e-6,
                                                       M, 3584, xn.data_ptr(), s)
                        assert rc >= 0, rc
                        return n
                    with torch.cuda.stream(st):
                        nsp = pair(0, st.cuda_stream); torch.cuda.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=st):
                            for i in range(a.pairs):
                                pair(i, st.cuda_stream)
                        g.replay(); torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(st)
                        for _ in range(a.reps):
                            g.replay()
                        e1.record(st); torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / (a.reps * a.pairs)
                    print(f"{name:5s} M={M:3d} nsplit={ns or 'auto':>4} (-> {nsp:2d}) cfg={cfg or 'auto':>4}: {us:7.2f} us per [GEMM + consumer] pair", flush=True)
    lib.mi355_debug_set(1, 0); lib.mi355_debug_set(2, 0)


if __name__ == "__main__":
    main()
