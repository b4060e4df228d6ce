#!/usr/bin/env python3
"""Micro-benchmark of the weight-only GEMM through the C-ABI (HBM-resident weights: several copies rotated
so the 256 MiB Infinity Cache cannot hold them).  usage: gemm_bench.py [--kinds w4,int8,fp16] [--ms 1,16,64]
[--var V] [--nsplit S] [--nbw B]"""
import argparse
import ctypes as C
import os
import sys

import torch

if "--product" not in sys.argv:
    os.environ["MI355_TUNING_LIB"] = "1"   # experiment switches live in the tuning build only (python -m rtp_llm_amd.build --tuning)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, model, ops  # noqa: E402

SHAPES = {"gu_k1792": (1792, 37888), "gu_k7168": (7168, 37888), "gu_k14336": (14336, 37888), "qkv": (3584, 4608), "o": (3584, 3584), "gate_up": (3584, 37888), "down": (18944, 3584), "lm_head": (3584, 152064)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="w4")
    ap.add_argument("--ms", default="1,16,64")
    ap.add_argument("--shapes", default="qkv,o,gate_up,down")
    ap.add_argument("--var", type=int, default=0)
    ap.add_argument("--nsplit", type=int, default=0)
    ap.add_argument("--nbw", type=int, default=0, help="block-shape cfg override + 1")
    ap.add_argument("--iters", type=int, default=48)
    ap.add_argument("--partial", type=int, default=0, help="1: qkv/o/down through mi355_linear_partial (split-K slabs, as the engine)")
    ap.add_argument("--nowide", type=int, default=0, help="1: disable the register-resident wide-M kernel")
    ap.add_argument("--nosmall", type=int, default=0, help="1: disable the persistent small-M kernel")
    ap.add_argument("--tune", default="", help="idx=val,... extra mi355_debug_set switches")
    ap.add_argument("--copies", type=int, default=0, help="weight copies rotated (0: enough to defeat the 256 MiB Infinity Cache; 1: cache-resident)")
    ap.add_argument("--product", action="store_true", help="time the product library (no experiment switches) instead of the tuning build")
    ap.add_argument("--bf16", type=int, default=0, help="1: bf16 activations (staged kernel, accumulator-side dequant)")
    a = ap.parse_args()
    adt = torch.bfloat16 if a.bf16 else torch.float16
    lib = _C.lib()
    if not a.product:
        lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]
        lib.mi355_debug_set(0, a.var); lib.mi355_debug_set(1, a.nsplit); lib.mi355_debug_set(2, a.nbw); lib.mi355_debug_set(4, a.nosmall); lib.mi355_debug_set(5, a.nowide)
        for kv_ in filter(None, a.tune.split(",")):
            lib.mi355_debug_set(int(kv_.split("=")[0]), int(kv_.split("=")[1]))
    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(0)
    for kind in a.kinds.split(","):
        for name in a.shapes.split(","):
            K, N = SHAPES[name]
            k = "fp16" if name == "lm_head" else kind
            base = model.synth_linear(K, N, k, dev, gen).pack(gate_up=(name == "gate_up" or name.startswith("gu_")), dtype=adt)
            ncopy = a.copies if a.copies > 0 else max(2, int(600e6 // base.nbytes) + 1)
            copies = [base] + [type(base)(base.qweight.clone(), None if base.meta is None else base.meta.clone(), base.wbits,
                                          base.K, base.N, base.K_pad, base.N_pad, base.group_size) for _ in range(ncopy - 1)]
            for M in [int(m) for m in a.ms.split(",")]:
                x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).to(adt)
                epi = _C.EPI_SILU_MUL if name in ("gate_up", "gu_k1792", "gu_k7168", "gu_k14336") else (_C.EPI_OUT_F32 if name == "lm_head" else 0)
                partial = a.partial and name not in ("gate_up", "lm_head")
                if partial:
                    slabs = torch.empty(16 * M * base.N_pad, dtype=torch.float32, device=dev)
                    structs = [ops.weight_struct(c, adt) for c in copies]
                    stream = torch.cuda.current_stream().cuda_stream
                    run = lambda i: lib.mi355_linear_partial(x.data_ptr(), M, C.byref(structs[i % ncopy]), slabs.data_ptr(), 16, stream)
                    ns = run(0)
                    assert ns > 0, _C.last_error() if hasattr(_C, "last_error") else ns
                else:
                    outs = [ops.linear(x, c, None, epi) for c in copies[:1]]
                    run = lambda i: ops.linear(x, copies[i % ncopy], None, epi, out=outs[0])
                torch.cuda.synchronize()
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for i in range(a.iters):
                    run(i)
                en.record(); torch.cuda.synchronize()
                us = st.elapsed_time(en) / a.iters * 1e3
                print(f"{k:5s} {name:8s} M={M:3d}  {us:8.2f} us  {base.nbytes / us / 1e3:8.1f} GB/s   ({base.nbytes / 1e6:.1f} MB, {ncopy} copies)", flush=True)


if __name__ == "__main__":
    main()
