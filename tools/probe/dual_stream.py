#!/usr/bin/env python3
"""Two half-batch engines (shared weights) replayed concurrently on two streams vs one full-batch engine: does overlapping the
ramp / drain of one chain with the steady state of the other beat reading the weights once?  usage: dual_stream.py [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtp_llm_amd import model
dev = "cuda:0"; B = int(sys.argv[1]) if len(sys.argv) > 1 else 64; ctx, page, steps = 1024, 16, 48
cfg = model.QWEN2_7B
gen = torch.Generator(device=dev).manual_seed(1)
layers = [model.synth_layer(cfg, "w4", dev, gen, zeros="centered") for _ in range(cfg.num_layers)]
weights = {"layers": layers, "embedding": (torch.randn(cfg.vocab, cfg.hidden, device=dev, generator=gen) * 0.5).half(),
           "final_norm": torch.ones(cfg.hidden, device=dev).half(), "lm_head": model.synth_linear(cfg.hidden, cfg.vocab, "fp16", dev, gen)}
def make(b, seed):
    msl = ctx + 4 * steps + 64; bps = (msl + page - 1) // page
    e = model.DecoderEngine(cfg, weights, kv_int8=False, page=page, num_blocks=b * bps, max_batch=b, max_seq_len=msl, device=dev)
    for kv in e.kv:
        kv.copy_(torch.randn(kv.shape, device=dev, dtype=torch.float16))
    ids = torch.randint(0, cfg.vocab, (b,), generator=torch.Generator().manual_seed(seed), dtype=torch.int32)
    bt = torch.randperm(b * bps, generator=torch.Generator().manual_seed(2)).reshape(b, bps).to(torch.int32)
    e.set_inputs(ids.tolist(), [ctx - 1] * b, bt); e.capture(b)
    return e
full = make(B, 1)
full.replay(B, 8); torch.cuda.synchronize()
t0 = time.perf_counter(); full.replay(B, steps); torch.cuda.synchronize(); t_full = (time.perf_counter() - t0) / steps
print(f"one engine  b={B}: {t_full * 1e3:.3f} ms/step  {B / t_full:.0f} tok/s")
h = B // 2
ea, eb = make(h, 3), make(h, 4)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(n):
    for _ in range(n):                      # interleave the submissions so that neither stream runs ahead on the host
        with torch.cuda.stream(sa): ea.replay(h, 1)
        with torch.cuda.stream(sb): eb.replay(h, 1)
run(8); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); t_dual = (time.perf_counter() - t0) / steps
print(f"two engines b={h} on two streams: {t_dual * 1e3:.3f} ms per pair of steps  {B / t_dual:.0f} tok/s")
with torch.cuda.stream(sa): ea.replay(h, 8)
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(sa): ea.replay(h, steps)
torch.cuda.synchronize(); t_half = (time.perf_counter() - t0) / steps
print(f"one engine  b={h}: {t_half * 1e3:.3f} ms/step")
