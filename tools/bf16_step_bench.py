#!/usr/bin/env python3
"""Decode step of Qwen2-7B W4A16 (synthetic weights) with bf16 activations / bf16 KV cache vs fp16, hipGraph replay, ctx 1024."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import model

dev = torch.device("cuda", 0)
cfg = model.MODELS["qwen2-7b"]
ctx, page = 1024, 16
for dtype in (torch.float16, torch.bfloat16):
    for B in (1, 8, 16, 32, 64):
        msl = ctx + 128
        bps = (msl + page - 1) // page
        gen = torch.Generator(device=dev).manual_seed(1)
        layers = [model.synth_layer(cfg, "w4", dev, gen, zeros="centered") for _ in range(cfg.num_layers)]
        w = {"layers": layers, "embedding": (torch.randn(cfg.vocab, cfg.hidden, device=dev, generator=gen) * 0.5).half(),
             "final_norm": torch.ones(cfg.hidden, device=dev).half(), "lm_head": model.synth_linear(cfg.hidden, cfg.vocab, "fp16", dev, gen)}
        eng = model.DecoderEngine(cfg, w, kv_int8=False, page=page, num_blocks=B * bps, max_batch=B, max_seq_len=msl, device=dev, dtype=dtype)
        del w, layers
        for kv in eng.kv:
            kv.copy_(torch.randn(kv.shape, device=dev, generator=gen, dtype=torch.float16).to(kv.dtype))
        bt = torch.randperm(B * bps, generator=torch.Generator().manual_seed(2)).reshape(B, bps).to(torch.int32)
        ids = torch.randint(0, cfg.vocab, (B,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
        eng.set_inputs(ids.tolist(), [ctx - 1] * B, bt)
        eng.capture(B)
        eng.replay(B, 4); torch.cuda.synchronize()
        eng.set_inputs(ids.tolist(), [ctx - 1] * B, bt)
        t0 = time.perf_counter(); eng.replay(B, 32); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 32 * 1e3
        prof = eng.profile(B, 2)
        print(f"{str(dtype):16s} b={B:3d}  {ms:7.3f} ms/step  {B / ms * 1e3:9.1f} tok/s   eager per class (ms/step): " +
              ", ".join(f"{k} {v['ms'] / 2:.3f}" for k, v in prof.items() if v['launches']), flush=True)
        del eng
        torch.cuda.empty_cache()
