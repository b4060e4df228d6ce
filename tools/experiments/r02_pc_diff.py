#!/usr/bin/env python3
"""Where does the producer/consumer GEMM differ from the wide kernel?  (tuning build; prints an error map by tile / row)"""
import ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtp_llm_amd import _C, model, ops
lib = _C.lib(); lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]
dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
K, N = 3584, 37888
w = model.synth_linear(K, N, "w4", dev, gen, zeros="centered").pack(gate_up=True)
for M in (33, 64):
    x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
    lib.mi355_debug_set(5, 2); ref = ops.linear(x, w, None, _C.EPI_SILU_MUL).float()
    lib.mi355_debug_set(5, 0); lib.mi355_debug_set(7, int(os.environ.get("PCDBG", "0")))
    for rep in range(3):
        y = ops.linear(x, w, None, _C.EPI_SILU_MUL).float(); torch.cuda.synchronize()
        d = (y - ref).abs()
        bad = d > 2e-2
        print(f"M={M} rep={rep} max={d.max().item():.4f} bad={int(bad.sum())}/{bad.numel()}")
        if bad.any():
            cols = bad.any(0).nonzero().flatten(); rows = bad.any(1).nonzero().flatten()
            tiles = torch.unique(cols // 16)
            print("  bad tiles:", tiles[:40].tolist(), "n=", len(tiles)); print("  bad rows:", rows.tolist()[:70])
            print("  tile%10 hist:", torch.bincount((tiles % 10), minlength=10).tolist())
