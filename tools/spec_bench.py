#!/usr/bin/env python3
"""Cost of one speculative round on the decode path (BASELINE config 5 shape: Qwen2-7B W4A16 target, Qwen2-0.5B fp16
draft, synthetic weights): draft gamma steps + target verify of B*(gamma+1) rows, against one plain decode step of B rows.
Acceptance is meaningless with random weights; the break-even acceptance follows from the two times."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import model
from rtp_llm_amd.speculative import SpeculativeDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--gamma", type=int, default=4)
ap.add_argument("--ctx", type=int, default=1024)
a = ap.parse_args()
dev = "cuda:0"
B, G, ctx, page = a.batch, a.gamma, a.ctx, 16
tc, dc = model.MODELS["qwen2-7b"], model.MODELS["qwen2-0.5b"]
nblk = B * ((ctx + 64 + page - 1) // page) + 8
mk = lambda cfg, kind, mb: model.DecoderEngine(cfg, model.synth_model(cfg, kind, dev, seed=1, zeros="centered"), kv_int8=False, page=page, num_blocks=nblk,
                                               max_batch=mb, max_seq_len=ctx + 64, device=dev)
target, draft = mk(tc, "w4", B * (G + 1)), mk(dc, "fp16", 2 * B)
bt = torch.arange(B * ((ctx + 64 + page - 1) // page), dtype=torch.int32).reshape(B, -1)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n


target.set_inputs([1] * B, [ctx] * B, bt)
t_plain = timed(lambda: target.forward(B))
rows = B * (G + 1)
target.set_inputs([1] * rows, [ctx + (r % (G + 1)) for r in range(rows)], bt[[r // (G + 1) for r in range(rows)]])
t_verify_indep = timed(lambda: target.forward(rows))                 # round-1 form: every verify row an independent decode row
target.set_inputs([1] * rows, [ctx + (r % (G + 1)) for r in range(rows)], bt)
t_verify = timed(lambda: target.forward(rows, q_len=G + 1))          # multi-row attention: a sequence's KV is streamed once
draft.set_inputs([1] * B, [ctx] * B, bt)
t_draft = timed(lambda: draft.forward(B))
spec = SpeculativeDecoder(target, draft, G)
spec.start([1] * B, [ctx] * B, bt, bt)
spec.step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    spec.step()
torch.cuda.synchronize()
t_round = (time.perf_counter() - t0) / 5 * 1e3
spec.profile = {}
for _ in range(5):
    spec.step()
print("phases (ms per round, synchronised):", {k: round(v / 5, 3) for k, v in spec.profile.items()})
dev_round = G * t_draft + t_verify
print(f"B={B} gamma={G} ctx={ctx}: target decode step (eager, {B} rows) {t_plain:.3f} ms; verify step ({rows} rows) {t_verify:.3f} ms "
      f"(= {t_verify / t_plain:.2f} x the plain step; as {rows} independent decode rows: {t_verify_indep:.3f} ms); draft step {t_draft:.3f} ms")
print(f"device time per round ~ {dev_round:.3f} ms (gamma draft steps + verify); python driver round {t_round:.3f} ms (host-bound, un-captured)")
print(f"break-even: {dev_round / t_plain:.2f} tokens per round (1 + accepted drafts) for speculative to match plain decoding")
