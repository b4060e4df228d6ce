#!/usr/bin/env python3
"""Does the row stride of x (K * 2 bytes) matter for kernels whose blocks all read the same activations from L2?
Runs the full-K o-proj and the wide gate_up at M = 64 for several K around 3584 (kernel-trace durations via ktrace.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtp_llm_amd import _C, model, ops
dev = "cuda:0"; gen = torch.Generator(device=dev).manual_seed(0)
for K in (3328, 3456, 3584, 3712, 3840, 4096):
    wo = model.synth_linear(K, 3584, "w4", dev, gen, zeros="centered").pack()
    wg = model.synth_linear(K, 37888, "w4", dev, gen, zeros="centered").pack(gate_up=True)
    x = (torch.randn(64, K, device=dev, generator=gen) * 0.5).half()
    res = torch.randn(64, 3584, device=dev, generator=gen).half()
    for _ in range(5):
        ops.linear_residual(x, wo, res)
        ops.linear(x, wg, None, _C.EPI_SILU_MUL)
    torch.cuda.synchronize()
