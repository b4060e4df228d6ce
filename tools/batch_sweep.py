#!/usr/bin/env python3
"""ms/step of the graph-replayed decode step at many batch heights (same engine, Qwen2-7B W4A16, ctx 1024): finds plan anomalies.
usage: batch_sweep.py [--batches 1,2,...] [--tune i=v,...]"""
import argparse, ctypes as C, os, sys, time
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,2,3,4,6,8,9,12,16,17,20,24,28,32,33,40,48,56,64"); ap.add_argument("--tune", default="")
ap.add_argument("--ctx", type=int, default=1024); ap.add_argument("--kv8", action="store_true")
a = ap.parse_args()
if a.tune: os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, model  # noqa: E402
lib = _C.lib()
for kv_ in filter(None, a.tune.split(",")):
    lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]; lib.mi355_debug_set(int(kv_.split("=")[0]), int(kv_.split("=")[1]))
dev = torch.device("cuda", 0); cfg = model.MODELS["qwen2-7b"]; page = 16; B = 64; ctx = a.ctx
max_seq = ctx + 256; bps = (max_seq + page - 1) // page
gen = torch.Generator(device=dev).manual_seed(1000)
layers = [model.synth_layer(cfg, "w4", dev, gen, zeros="centered") for _ in range(cfg.num_layers)]
w = {"layers": layers, "embedding": (torch.randn(cfg.vocab, cfg.hidden, device=dev, generator=gen) * 0.5).half(),
     "final_norm": torch.ones(cfg.hidden, device=dev).half(), "lm_head": model.synth_linear(cfg.hidden, cfg.vocab, "fp16", dev, gen)}
eng = model.DecoderEngine(cfg, w, kv_int8=a.kv8, page=page, num_blocks=B * bps, max_batch=B, max_seq_len=max_seq, device=dev)
del w, layers; torch.cuda.empty_cache()
for l in range(cfg.num_layers):
    if a.kv8:
        eng.kv[l].copy_(torch.randint(-127, 128, eng.kv[l].shape, device=dev, dtype=torch.int8)); eng.kv_scale[l].fill_(0.01)
    else:
        eng.kv[l].copy_(torch.randn(eng.kv[l].shape, device=dev, dtype=torch.float16))
bt = torch.randperm(B * bps, generator=torch.Generator().manual_seed(2)).reshape(B, bps).to(torch.int32)
ids = torch.randint(0, cfg.vocab, (B,), dtype=torch.int32)
for b in [int(v) for v in a.batches.split(",")]:
    eng.set_inputs(ids.tolist(), [ctx - 1] * B, bt)
    eng.capture(b); eng.replay(b, 4); torch.cuda.synchronize()
    eng.set_inputs(ids.tolist(), [ctx - 1] * B, bt)
    t0 = time.perf_counter(); eng.replay(b, 32); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 32 * 1e3
    print(f"b={b:3d}  {ms:7.4f} ms/step  {b / ms * 1e3:9.1f} tok/s", flush=True)
