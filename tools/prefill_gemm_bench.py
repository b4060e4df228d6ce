#!/usr/bin/env python3
"""Throughput of the large-M weight-only GEMM (gemm_prefill.hip) at the Qwen2-7B layer shapes.
usage: prefill_gemm_bench.py [--ms 128,512,2048,4096] [--kind w4]"""
import argparse, ctypes, os, sys
import torch
if "--tune" in " ".join(sys.argv):
    os.environ["MI355_TUNING_LIB"] = "1"   # experiment switches live in the tuning build only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, model, ops

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="128,512,2048,4096"); ap.add_argument("--kind", default="w4"); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--tune", default="", help="idx=val,... forwarded to mi355_debug_set (3=2 / 3=4: tiles per wave of the prefill kernel)")
ap.add_argument("--shapes", default="qkv,o,gate_up,down")
a = ap.parse_args()
for kv_ in filter(None, a.tune.split(",")):
    _C.lib().mi355_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
    _C.lib().mi355_debug_set(int(kv_.split("=")[0]), int(kv_.split("=")[1]))
dev = "cuda:0"
SH = {"qkv": (3584, 4608), "o": (3584, 3584), "gate_up": (3584, 37888), "down": (18944, 3584)}
gen = torch.Generator(device=dev).manual_seed(0)
for name, (K, N) in [(n, SH[n]) for n in a.shapes.split(",")]:
    w = model.synth_linear(K, N, a.kind, dev, gen, zeros="centered").pack(gate_up=(name == "gate_up"))
    for M in [int(v) for v in a.ms.split(",")]:
        x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
        epi = _C.EPI_SILU_MUL if name == "gate_up" else _C.EPI_NONE
        out = ops.linear(x, w, epilogue=epi)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.linear(x, w, epilogue=epi, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        print(f"{name:8s} M={M:5d} K={K:5d} N={N:5d}  {us:9.1f} us  {2.0 * M * K * N / us / 1e6:8.1f} TFLOP/s", flush=True)
