#!/usr/bin/env python3
"""Where does the engine deviate from the oracle at full width?  Runs the C++ step driver in its segmented (tp-style)
mode with UNSPLIT weights and no collective, so the o-proj / down-proj outputs of every layer are visible in ar_buf, and
compares each with an instrumented copy of the oracle's layer loop fed the SAME inputs.
usage: parity_probe.py [--batch 64 --ctx 1024 --kv-int8 --kind w4]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from rtp_llm_amd import _C, kvcache, model

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--ctx", type=int, default=1024)
ap.add_argument("--kv-int8", action="store_true"); ap.add_argument("--kind", default="w4"); ap.add_argument("--steps", type=int, default=2); ap.add_argument("--zeros", default="centered")
a = ap.parse_args()
DEV = "cuda:0"
B, ctx, kv_int8 = a.batch, a.ctx, a.kv_int8
cfg = model.ModelConfig("qwen2-7b-2l", 2, 3584, 28, 4, 128, 18944, 152064, max_pos=ctx + 16)
w_dev = model.synth_model(cfg, a.kind, DEV, seed=21, zeros=a.zeros)
w = model.weights_to(w_dev, "cpu")
dense = lambda c: (c.w.float() if c.kind == "fp16" else oracle.dequant_int8(c.q, c.scales) if c.kind == "int8"
                   else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size))
page = 16
mb = (ctx + a.steps + page - 1) // page
eng = model.DecoderEngine(cfg, w_dev, kv_int8=kv_int8, page=page, num_blocks=B * mb, max_batch=B, max_seq_len=ctx + a.steps, device=DEV, tp_size=2)
Wd = [{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")} for L in w["layers"]]
lm = dense(w["lm_head"])
okv = oracle.OracleKV(cfg.num_layers, B, kv_int8)
g = torch.Generator().manual_seed(5)
bt = torch.randperm(B * mb, generator=g).reshape(B, mb).to(torch.int32)
for l in range(cfg.num_layers):
    for b in range(B):
        K = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=g).half(); V = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=g).half()
        if kv_int8:
            Kq, ks = oracle.quant_kv_int8(K); Vq, vs = oracle.quant_kv_int8(V)
            kvcache.write_tokens(eng.kv[l], eng.kv_scale[l], bt[b], 0, Kq, Vq, ks, vs)
            okv.k[l][b], okv.v[l][b], okv.ks[l][b], okv.vs[l][b] = list(Kq), list(Vq), list(ks), list(vs)
        else:
            kvcache.write_tokens(eng.kv[l], None, bt[b], 0, K, V)
            okv.k[l][b], okv.v[l][b] = list(K), list(V)
tok = torch.randint(0, cfg.vocab, (B,), generator=g, dtype=torch.int32)
cs = oracle.rope_cos_sin(cfg.hd, cfg.rope_theta, cfg.max_pos)
nh, nkv, hd, eps = cfg.nh, cfg.nkv, cfg.hd, cfg.rms_eps
st = torch.cuda.current_stream().cuda_stream

def rep(name, got, ref):
    d = (got.float() - ref.float()).abs()
    lim = 1e-2 + 1e-2 * ref.float().abs()
    print(f"  {name:28s} max|d| {d.max():.5f}  |ref|max {ref.float().abs().max():.3f}  rms {ref.float().pow(2).mean().sqrt():.4f}  "
          f"beyond-tol {(d > lim).float().mean():.2e}  worst row {int(d.max(dim=-1).values.argmax())}")

for step in range(a.steps):
    print(f"step {step}")
    pos = torch.full((B,), ctx - 1 + step, dtype=torch.int32)
    eng.set_inputs(tok.tolist(), pos.tolist(), bt)
    _C.check(eng.lib.mi355_decoder_begin(eng.handle, B, st))
    h = w["embedding"][tok.long()]
    for l in range(cfg.num_layers):
        L = w["layers"][l]
        # --- oracle attention half on the oracle's own stream h
        x = oracle.rmsnorm(h, L["input_norm"], eps)
        qkv = oracle.linear(x, Wd[l]["qkv"], L["qkv_bias"])
        qh = oracle.apply_rope(qkv[:, :nh * hd].reshape(B, nh, hd), pos, cs)
        kh = oracle.apply_rope(qkv[:, nh * hd:(nh + nkv) * hd].reshape(B, nkv, hd), pos, cs)
        vh = qkv[:, (nh + nkv) * hd:].reshape(B, nkv, hd)
        attn = torch.empty(B, nh * hd, dtype=h.dtype)
        for b in range(B):
            okv.append(l, b, kh[b], vh[b])
            K, V, ks, vs = okv.get(l, b)
            attn[b] = oracle.attention_decode(qh[b], K, V, 1 / math.sqrt(hd), ks, vs).reshape(-1)
        o = oracle.linear(attn, Wd[l]["o"])
        _C.check(eng.lib.mi355_decoder_layer_attn(eng.handle, l, st)); torch.cuda.synchronize()
        rep(f"L{l} o-proj out", eng.ar_buf[:B].cpu(), o)
        # feed the ORACLE's value forward so that errors do not accumulate across segments
        eng.ar_buf[:B].copy_(o)
        h = h + o
        x = oracle.rmsnorm(h, L["post_norm"], eps)
        act = oracle.silu_mul(oracle.linear(x, Wd[l]["gate_up"]))
        dn = oracle.linear(act, Wd[l]["down"])
        _C.check(eng.lib.mi355_decoder_layer_mlp(eng.handle, l, st)); torch.cuda.synchronize()
        rep(f"L{l} down out", eng.ar_buf[:B].cpu(), dn)
        eng.ar_buf[:B].copy_(dn)
        h = h + dn
    hn = oracle.rmsnorm(h, w["final_norm"], eps)
    logits = oracle.linear(hn, lm, out_f32=True)
    _C.check(eng.lib.mi355_decoder_finish(eng.handle, 0, st)); torch.cuda.synchronize()
    rep("final hidden", eng.hidden[:B].cpu(), hn)
    rep("logits", eng.logits[:B].cpu(), logits)
    tok = oracle.greedy(logits)
