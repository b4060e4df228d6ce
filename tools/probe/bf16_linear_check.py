"""bf16 linear (staged kernel, accumulator-side W4 dequant, bf16 MFMA) against the oracle with bf16 tensors."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from rtp_llm_amd import _C, model, ops
DEV = "cuda:0"
g = lambda s: torch.Generator().manual_seed(s)
dense = lambda c: c.w.to(torch.bfloat16).float() if c.kind == "fp16" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size)
worst = 0.0
for kind, group in (("w4", 128), ("w4", 64), ("w4", 32), ("fp16", 0)):
    for K, N in ((256, 64), (1024, 4608), (3584, 512), (9472, 896)):
        c = model.synth_linear(K, N, kind, "cpu", g(10 + K + N), group or 128)
        for M in (1, 7, 16, 33, 64, 83, 200):
            x = (torch.randn(M, K, generator=g(M)) * 0.5).to(torch.bfloat16)
            bias = (torch.randn(N, generator=g(3)) * 0.1).to(torch.bfloat16) if N % 3 == 0 else None
            ref = oracle.linear(x, dense(c), bias)
            for epi in (_C.EPI_NONE, _C.EPI_OUT_F32):
                y = ops.linear(x.to(DEV), c.pack(dtype=torch.bfloat16).to(DEV), None if bias is None else bias.to(DEV), epi)
                torch.cuda.synchronize()
                r = oracle.linear(x, dense(c), bias, out_f32=True) if epi else ref
                err = float((y.cpu().float() - r.float()).abs().max()); scale = float(r.float().abs().max())
                worst = max(worst, err / scale)
                ok = torch.allclose(y.cpu().float(), r.float(), atol=2e-2, rtol=2e-2)
                if not ok or M in (1, 64):
                    print(f"{kind} g{group} K={K} N={N} M={M} epi={epi}: max err {err:.3e} (max |ref| {scale:.2f}) {'ok' if ok else 'FAIL'}", flush=True)
print("worst relative-to-range error", worst)
