#!/usr/bin/env python3
"""gate_up at a few rows: fused-norm full-K launch vs staged split-K + fold (kernel durations via ktrace.sh)."""
import ctypes as C, os, sys
import torch
TUNING = os.environ.get("PROBE_TUNING", "0") == "1"
if TUNING:
    os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtp_llm_amd import _C, model, ops
lib = _C.lib()
if TUNING:
    lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]
dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
H, I = 3584, 18944
ws = [model.synth_linear(H, 2 * I, "w4", dev, gen, zeros="centered").pack(gate_up=True) for _ in range(6)]
gamma = torch.ones(H, dtype=torch.float16, device=dev)
for M in (1, 8):
    h = torch.randn(M, H, device=dev, generator=gen).half()
    ssq = torch.zeros(16, H // 16, dtype=torch.float32, device=dev); ssq[:M] = (h.float() ** 2).reshape(M, H // 16, 16).sum(-1)
    for dbg in ((0, 1) if TUNING else (0,)):
        if TUNING:
            lib.mi355_debug_set(6, dbg)
        for i in range(12):
            ops.norm_linear(h, (ssq, gamma, 1e-6), ws[i % 6], None, _C.EPI_SILU_MUL)
        torch.cuda.synchronize()
    if TUNING:
        lib.mi355_debug_set(6, 0)
    for i in range(12):
        ops.linear(h, ws[i % 6], None, _C.EPI_SILU_MUL | _C.HINT_NO_PERSISTENT)
    torch.cuda.synchronize()
