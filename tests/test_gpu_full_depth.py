"""Parity at FULL DEPTH: the object the bench line measures -- Qwen2-7B, all 28 layers, W4 g128, ctx 1024 -- against the CPU oracle.

Every other end-to-end test runs 1-3 layers.  The step of round 4 carries a depth-sensitive mechanism (the post-attention RMSNorm is
deferred into gate_up: gamma 2^-e h' travels as an fp16 image, csrc/engine.cpp decoder_create), and rounding differences between
the HIP kernels and the oracle's fp32 matmuls accumulate along the residual stream, so this file checks what 2 layers cannot:

  * fp16, 16-bit cache: B = 64 (the headline configuration: image launches, 6 per layer), B = 8 (one row block) and B = 1 (the
    few-row full-K launches), two greedy steps each, hipGraph-replayed; logits within 1e-2 (north_star), greedy ids identical on
    every row whose top-2 margin exceeds the tolerance;
  * the first step of each also runs EAGERLY through the segment calls (mi355_decoder_begin / layer_attn / layer_mlp), and the
    residual stream after every half layer is compared with the oracle's: the per-layer max |delta h| shows where a drift starts;
    max |h'| is recorded against the fp16 range the deferred-norm image has (|gamma 2^-e h'| <= |h'|);
  * one bf16 step at B = 64 (bf16 KV cache);
  * configs[1] at full depth: per-channel W8 (load-time INT8 autoquant), B = 16, two steps on the W8 image launches of round 5.

Reference: Qwen3Model.forward / Qwen3DecoderLayer.forward (rtp_llm/models_py/model_desc/qwen3.py:57-79,124-138), the generate loop of
standalone/auto_model.py:144-265.  The oracle dequantises ONE layer at a time (932 MB fp32) so the host never holds the 26 GB of a
dense 28-layer model.  Results go to gpurun_out/full_depth_parity.json (committed copy: profiles/r05_full_depth_parity.json).
"""
import json
import math
import os

import pytest
import torch

from oracle import oracle
from rtp_llm_amd import _C, kvcache, model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CTX, PAGE, STEPS = 1024, 16, 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RECORD = {}


@pytest.fixture(scope="module", autouse=True)
def _require_native():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _C.lib()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    yield
    if _RECORD:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "full_depth_parity.json"), "w") as f:
            json.dump(_RECORD, f, indent=1)


class _LazyLayers:
    """w["layers"] of OracleDecoder: layer l is dequantised when asked for, and only the last one is kept."""

    def __init__(self, layers, cast=lambda t: t):
        self.layers, self.cast, self.at, self.cur = layers, cast, -1, None

    def __getitem__(self, l):
        if l != self.at:
            L = self.layers[l]
            self.cur = None
            dq = lambda c: oracle.dequant_int8(c.q, c.scales) if c.kind == "int8" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size)
            self.cur = {"input_norm": self.cast(L["input_norm"]), "post_norm": self.cast(L["post_norm"]), "qkv_bias": self.cast(L["qkv_bias"]),
                        **{k: dq(L[k]) for k in ("qkv", "o", "gate_up", "down")}}
            self.at = l
        return self.cur


class _TensorKV:
    """OracleKV's contract (append / get) on one tensor per (layer, sequence) instead of a Python list of rows: 28 layers x 64
    sequences x 1024 tokens would be 1.8 M list entries."""
    int8 = False

    def __init__(self, K, V, n):
        self.K, self.V, self.n = K, V, n              # [L][B] tensors [cap, nkv, hd], token counts

    @classmethod
    def fill(cls, num_layers, B, cap, ctx, nkv, hd, dtype, gen):
        K = [[torch.zeros(cap, nkv, hd, dtype=dtype) for _ in range(B)] for _ in range(num_layers)]
        V = [[torch.zeros(cap, nkv, hd, dtype=dtype) for _ in range(B)] for _ in range(num_layers)]
        for l in range(num_layers):
            for b in range(B):
                K[l][b][:ctx] = torch.randn(ctx, nkv, hd, generator=gen).to(dtype)
                V[l][b][:ctx] = torch.randn(ctx, nkv, hd, generator=gen).to(dtype)
        return cls(K, V, [[ctx] * B for _ in range(num_layers)])

    def fork(self, B):
        """The first B sequences at their initial length (appends of an earlier run are overwritten, never read)."""
        return _TensorKV(self.K, self.V, [[CTX - 1] * B for _ in self.n])

    def append(self, layer, b, k, v):
        i = self.n[layer][b]
        self.K[layer][b][i], self.V[layer][b][i] = k, v
        self.n[layer][b] = i + 1

    def get(self, layer, b):
        i = self.n[layer][b]
        return self.K[layer][b][:i], self.V[layer][b][:i], None, None


@pytest.fixture(scope="module")
def full_model():
    cfg = model.ModelConfig("qwen2-7b", 28, 3584, 28, 4, 128, 18944, 152064, max_pos=CTX + 16)
    w_dev = model.synth_model(cfg, "w4", DEV, seed=28, zeros="centered")
    w = model.weights_to(w_dev, "cpu")
    return cfg, w_dev, w


def _engine(cfg, w_dev, B, dtype):
    mb = (CTX + STEPS + PAGE - 1) // PAGE
    return model.DecoderEngine(cfg, w_dev, kv_int8=False, page=PAGE, num_blocks=B * mb, max_batch=B, max_seq_len=CTX + STEPS, device=DEV,
                               dtype=dtype), mb


def _resid(eng, B):
    """The residual stream of the step in flight: the first buffer carved out of the engine's workspace (csrc/engine.cpp carve_all)."""
    H = eng.cfg.hidden
    return eng.workspace[: eng.max_batch * H * 2].view(eng.dtype).view(eng.max_batch, H)[:B].float().cpu()


def _eager_trace(eng, B):
    """One step through the segment calls, the residual stream read back after every half layer (no sampling, nothing advances)."""
    st, h, lib = eng._st(), eng.handle, eng.lib
    out = []
    _C.check(lib.mi355_decoder_begin(h, B, st), "decoder_begin")
    for l in range(eng.cfg.num_layers):
        _C.check(lib.mi355_decoder_layer_attn(h, l, st), "decoder_layer_attn")
        torch.cuda.synchronize()
        out.append(_resid(eng, B))
        _C.check(lib.mi355_decoder_layer_mlp(h, l, st), "decoder_layer_mlp")
        torch.cuda.synchronize()
        out.append(_resid(eng, B))
    _C.check(lib.mi355_decoder_finish(h, 0, st), "decoder_finish")
    torch.cuda.synchronize()
    return out


def _run(tag, cfg, eng, odec, okv, bt, B, steps, parity, tol=1e-2):
    g = torch.Generator().manual_seed(100 + B)
    tok = torch.randint(0, cfg.vocab, (B,), generator=g, dtype=torch.int32)
    eng.set_inputs(tok.tolist(), [CTX - 1] * B, bt[:B])
    eng.capture(B)
    rec = {"rows": B, "steps": [], "per_layer": []}
    for step in range(steps):
        pos = torch.full((B,), CTX - 1 + step, dtype=torch.int32)
        trace = [] if step == 0 else None
        _, ref_logits = odec.forward_tokens(tok, pos, okv, list(range(B)), trace=trace)
        if step == 0:   # where along the depth do kernel and oracle part?  (eager segments; the replay below repeats the same step)
            got_h = _eager_trace(eng, B)
            for (l, half, ref_h), gh in zip(trace, got_h):
                d = (gh - ref_h.float()).abs()
                rec["per_layer"].append({"layer": l, "after": half, "max_abs_dh": float(d.max()), "mean_abs_dh": float(d.mean()),
                                         "max_abs_h": float(ref_h.float().abs().max()), "rms_h": float(ref_h.float().pow(2).mean().sqrt())})
        eng.replay(B, 1)
        torch.cuda.synchronize()
        got = eng.logits[:B].cpu()
        err = float((got - ref_logits).abs().max())
        ref_next = oracle.greedy(ref_logits)
        got_next = eng.token_ids[:B].cpu()
        r = parity.step(got_ids=got_next, ref_ids=ref_next, ref_logits=ref_logits, got_logits=got, tol=tol, label=f"{tag} step {step}")
        rec["steps"].append({k: r[k] for k in ("rows", "exact", "safe", "max_abs_logit_err", "min_top2_margin")})
        _RECORD[tag] = rec
        assert torch.allclose(got, ref_logits, atol=tol, rtol=tol), (tag, step, err)
        tok = ref_next
        eng.token_ids[:B].copy_(tok)
    # the image of the deferred norm holds gamma 2^-e h' with |gamma 2^-e| <= 1: its headroom is the fp16 range over max |h'|
    hmax = max(p["max_abs_h"] for p in rec["per_layer"])
    rec["max_abs_h_over_depth"] = hmax
    rec["fp16_image_headroom"] = 65504.0 / hmax
    rec["max_abs_dh_last_layer"] = rec["per_layer"][-1]["max_abs_dh"]
    _RECORD[tag] = rec
    assert eng.oob_count() == 0
    return rec


def test_full_depth_fp16_b64_b8_b1_vs_oracle(full_model, parity):
    cfg, w_dev, w = full_model
    eng, mb = _engine(cfg, w_dev, 64, torch.float16)
    odec = oracle.OracleDecoder({**cfg.__dict__}, {"embedding": w["embedding"], "final_norm": w["final_norm"],
                                                   "lm_head": w["lm_head"].w.float(), "layers": _LazyLayers(w["layers"])})
    g = torch.Generator().manual_seed(7)
    bt = torch.randperm(64 * mb, generator=g).reshape(64, mb).to(torch.int32)
    base = _TensorKV.fill(cfg.num_layers, 64, CTX + STEPS, CTX - 1, cfg.nkv, cfg.hd, torch.float16, g)
    for l in range(cfg.num_layers):
        for b in range(64):
            kvcache.write_tokens(eng.kv[l], None, bt[b], 0, base.K[l][b][:CTX - 1], base.V[l][b][:CTX - 1])
    for B in (64, 8, 1):
        rec = _run(f"fp16-b{B}", cfg, eng, odec, base.fork(B), bt, B, STEPS, parity)
        assert rec["fp16_image_headroom"] > 4.0, rec["max_abs_h_over_depth"]


def test_full_depth_bf16_b64_vs_oracle(full_model, parity):
    BF = torch.bfloat16
    cfg, w_dev, w = full_model
    eng, mb = _engine(cfg, w_dev, 64, BF)
    bf = lambda t: None if t is None else t.to(BF)
    odec = oracle.OracleDecoder({**cfg.__dict__}, {"embedding": bf(w["embedding"]), "final_norm": bf(w["final_norm"]),
                                                   "lm_head": w["lm_head"].w.to(BF).float(), "layers": _LazyLayers(w["layers"], bf)})
    g = torch.Generator().manual_seed(8)
    bt = torch.randperm(64 * mb, generator=g).reshape(64, mb).to(torch.int32)
    base = _TensorKV.fill(cfg.num_layers, 64, CTX + STEPS, CTX - 1, cfg.nkv, cfg.hd, BF, g)
    for l in range(cfg.num_layers):
        for b in range(64):
            kvcache.write_tokens(eng.kv[l], None, bt[b], 0, base.K[l][b][:CTX - 1], base.V[l][b][:CTX - 1])
    # bf16 keeps 8 significant bits: one ulp of a logit of magnitude 2-4 is 0.016, and 28 layers of one-ulp flips between the
    # kernels' and the oracle's summation orders reach the lm_head -- 2.9e-2 measured (fp16 at the same depth: 3.7e-3), so the step is
    # held to the 3e-2 of the bf16 tests (tests/test_gpu_bf16.py), greedy ids identical on every row whose top-2 margin exceeds it
    _run("bf16-b64", cfg, eng, odec, base.fork(64), bt, 64, 1, parity, tol=3e-2)


def test_full_depth_w8a16_b16_vs_oracle(parity):
    """BASELINE configs[1] at full depth: Qwen2-7B with load-time INT8 autoquant (per-channel W8, device_impl.py:183-222), 28 layers, B = 16,
    ctx 1024, fp16 cache -- the W8 instances of the image launches (gemm_fullk64 / gemm_wide / gemm_splitk64) over the whole depth."""
    cfg = model.ModelConfig("qwen2-7b", 28, 3584, 28, 4, 128, 18944, 152064, max_pos=CTX + 16)
    w_dev = model.synth_model(cfg, "int8", DEV, seed=29)
    w = model.weights_to(w_dev, "cpu")
    B = 16
    eng, mb = _engine(cfg, w_dev, B, torch.float16)
    del w_dev
    odec = oracle.OracleDecoder({**cfg.__dict__}, {"embedding": w["embedding"], "final_norm": w["final_norm"],
                                                   "lm_head": w["lm_head"].w.float(), "layers": _LazyLayers(w["layers"])})
    g = torch.Generator().manual_seed(9)
    bt = torch.randperm(B * mb, generator=g).reshape(B, mb).to(torch.int32)
    base = _TensorKV.fill(cfg.num_layers, B, CTX + STEPS, CTX - 1, cfg.nkv, cfg.hd, torch.float16, g)
    for l in range(cfg.num_layers):
        for b in range(B):
            kvcache.write_tokens(eng.kv[l], None, bt[b], 0, base.K[l][b][:CTX - 1], base.V[l][b][:CTX - 1])
    _run("w8a16-b16", cfg, eng, odec, base.fork(B), bt, B, STEPS, parity)
