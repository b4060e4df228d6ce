#!/usr/bin/env python3
"""Launch times of the sampler kernels behind lm_head at the Qwen2 vocabulary (B rows x 152064 fp32 logits, 2048-token history)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import ops  # noqa: E402


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64); ap.add_argument("--vocab", type=int, default=152064); ap.add_argument("--hist", type=int, default=2048)
    a = ap.parse_args()
    dev = "cuda:0"
    B, V, L = a.batch, a.vocab, a.hist
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(B, V, generator=g) * 3).to(dev)
    hist = torch.randint(0, V, (L, B), generator=g, dtype=torch.int32).to(dev)
    ones = torch.ones(B)
    probs = ops.softmax_rows(logits)
    u = torch.rand(B, generator=g).to(dev)
    rows = [
        ("argmax (greedy fast path)", lambda: ops.argmax(logits)),
        ("apply_penalties: temperature", lambda: ops.apply_penalties(logits, temperature=ones * 0.8)),
        ("apply_penalties: repetition + presence + frequency", lambda: ops.apply_penalties(logits, repetition_penalty=ones * 1.1, presence_penalty=ones * 0.1,
                                                                                              frequency_penalty=ones * 0.1, output_ids=hist, max_input_length=0, step=L)),
        ("ban_repeat_ngram (n = 3)", lambda: ops.ban_repeat_ngram(logits, hist.t().contiguous(), torch.full((B,), L - 1, dtype=torch.int32), torch.full((B,), 3, dtype=torch.int32))),
        ("softmax_rows", lambda: ops.softmax_rows(logits)),
        ("sample_rows (no filter)", lambda: ops.sample_rows(probs, u)),
        ("top_k_top_p_sample (k = 50, p = 0.9)", lambda: ops.top_k_top_p_sample(probs, torch.full((B,), 50, dtype=torch.int32), ones * 0.9, u)),
        ("top_k_top_p_sample (p = 0.9, renormalised probabilities out)", lambda: ops.top_k_top_p_sample(probs, None, ones * 0.9, u, return_probs=True)),
    ]
    print(f"# sampler kernels, B = {B} rows x V = {V} fp32, history {L} tokens (host wrapper + launch, us per call)")
    for name, fn in rows:
        print(f"{name:62s} {timed(fn):9.1f} us")


if __name__ == "__main__":
    main()
