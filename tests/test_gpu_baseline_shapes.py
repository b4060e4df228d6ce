"""GPU parity at the BASELINE.json shapes, against the CPU oracle directly (no property stand-ins).

  * every linear of Qwen2-7B and Llama-3-70B (W4 g128; Qwen2-7B also W8 per-channel) at the batch heights where the
    kernel selection changes (M = 1, 8, 16, 17, 32, 33, 48, 64): HIP GEMM vs oracle.linear on the dequantised weights;
  * a full-width 2-layer Qwen2-7B engine step, hipGraph-replayed, against oracle.OracleDecoder:
      - W4 g128, B = 64, ctx 1024, fp16 KV        (the configuration BASELINE.json's metric is quoted on)
      - W4 g128, B = 64, ctx 4096, INT8 KV        (configs[2])
      - W8 per-channel, B = 16, ctx 1024, fp16 KV (configs[1])
      - W4 g128, B = 1 and B = 8, ctx 1024        (the small-batch step: full-K launches, RMSNorm on load, 6 launches per layer)
    logits within 1e-2 (north_star's tolerance), greedy ids identical wherever the oracle's top-2 margin exceeds it.

The oracle computes M = 64 rows once per weight and every smaller M is checked against its leading rows (rows of a GEMM
are independent), so the CPU side stays at a few seconds per shape.
"""
import math

import pytest
import torch

from oracle import oracle
from rtp_llm_amd import _C, kvcache, model, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(atol=1e-2, rtol=1e-2)
MS = (1, 8, 16, 17, 32, 33, 48, 64)


@pytest.fixture(scope="module", autouse=True)
def _require_native():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _C.lib()
    torch.set_num_threads(min(32, torch.get_num_threads()))


def _dense(c):
    if c.kind == "fp16":
        return c.w.float()
    if c.kind == "int8":
        return oracle.dequant_int8(c.q, c.scales)
    return oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size)


def _shapes(cfg):
    qkv = (cfg.nh + 2 * cfg.nkv) * cfg.hd
    return {"qkv": (cfg.hidden, qkv), "o": (cfg.nh * cfg.hd, cfg.hidden), "gate_up": (cfg.hidden, 2 * cfg.inter),
            "down": (cfg.inter, cfg.hidden)}


LINEAR_CASES = [(m, k, name) for m, kinds in ((model.QWEN2_7B, ("w4", "int8")), (model.LLAMA3_70B, ("w4",)))
                for k in kinds for name in ("qkv", "o", "gate_up", "down")]


@pytest.mark.parametrize("cfg,kind,name", LINEAR_CASES, ids=[f"{m.name}-{k}-{n}" for m, k, n in LINEAR_CASES])
def test_linear_baseline_shape_vs_oracle(cfg, kind, name):
    K, N = _shapes(cfg)[name]
    gen = torch.Generator(device=DEV).manual_seed(K * 7 + N)
    c_dev = model.synth_linear(K, N, kind, DEV, gen)
    packed = c_dev.pack(gate_up=(name == "gate_up"))
    c = model.weights_to({"w": c_dev}, "cpu")["w"]
    W = _dense(c)
    del c
    x = (torch.randn(64, K, generator=torch.Generator().manual_seed(3)) * 0.5).half()
    bias = (torch.randn(N, generator=torch.Generator().manual_seed(4)) * 0.1).half() if name == "qkv" else None
    ref = oracle.linear(x, W, bias)
    if name == "gate_up":
        ref = oracle.silu_mul(ref)
    del W
    xd, bd = x.to(DEV), None if bias is None else bias.to(DEV)
    for M in MS:
        y = ops.linear(xd[:M].contiguous(), packed, bd, epilogue=_C.EPI_SILU_MUL if name == "gate_up" else _C.EPI_NONE)
        torch.cuda.synchronize()
        err = (y.cpu().float() - ref[:M].float()).abs().max()
        assert torch.allclose(y.cpu().float(), ref[:M].float(), **TOL), f"{name} M={M}: max err {err}"


# ------------------------------------------------------------------ per-rank shapes of configs[3] / configs[4] (only timed, never compared)
def _rank_shapes(cfg, tp):
    """Megatron split of one layer over tp ranks (reference: rtp_llm/utils/model_weight.py:447-509,1517-1563): column-parallel
    QKV / gate_up, row-parallel O / down; the intermediate size padded to tp x 128 as model.ModelConfig.padded_inter does."""
    inter = cfg.padded_inter(tp) // tp if hasattr(cfg, "padded_inter") else cfg.inter // tp
    nh, nkv = cfg.nh // tp, max(1, cfg.nkv // tp)
    return {"qkv": (cfg.hidden, (nh + 2 * nkv) * cfg.hd), "o": (nh * cfg.hd, cfg.hidden), "gate_up": (cfg.hidden, 2 * inter),
            "down": (inter, cfg.hidden)}


RANK_CASES = [(model.LLAMA3_70B, 8, 32, n) for n in ("qkv", "o", "gate_up", "down")] + \
             [(model.MODELS["qwen2-72b"], 8, 40, n) for n in ("qkv", "o", "gate_up", "down")]     # 72B: the 40-row verify step of configs[4] (8 x (gamma + 1))


@pytest.mark.parametrize("cfg,tp,M,name", RANK_CASES, ids=[f"{c.name}-tp{t}-m{m}-{n}" for c, t, m, n in RANK_CASES])
def test_linear_per_rank_shapes_vs_oracle(cfg, tp, M, name):
    """One rank's shard of every linear of Llama-3-70B (TP 8, 32 rows) and Qwen2-72B (TP 8, 40 rows of a verify step) against
    oracle.linear: the composed launch (mi355_linear_forward), the split-K slab entry the TP step uses for the row-parallel
    O / down, and -- where their plans take the shape -- the launches on activation images."""
    import ctypes as C
    K, N = _rank_shapes(cfg, tp)[name]
    gen = torch.Generator(device=DEV).manual_seed(K * 3 + N + tp)
    c_dev = model.synth_linear(K, N, "w4", DEV, gen)
    packed = c_dev.pack(gate_up=(name == "gate_up"))
    W = _dense(model.weights_to({"w": c_dev}, "cpu")["w"])
    x = (torch.randn(M, K, generator=torch.Generator().manual_seed(3)) * 0.5).half()
    ref32 = x.float() @ W
    ref = oracle.linear(x, W, None)
    xd = x.to(DEV)
    y = ops.linear(xd, packed, None, epilogue=_C.EPI_SILU_MUL if name == "gate_up" else _C.EPI_NONE)
    want = oracle.silu_mul(ref) if name == "gate_up" else ref
    assert torch.allclose(y.cpu().float(), want.float(), **TOL), f"{name}: {(y.cpu().float() - want.float()).abs().max()}"
    if name in ("o", "down"):      # row-parallel: partial sums leave as fp32 slabs for the fused all-reduce
        ws = ops.weight_struct(packed)
        slabs = torch.full((16, M, packed.N_pad), float("nan"), dtype=torch.float32, device=DEV)
        ns = _C.lib().mi355_linear_partial(xd.data_ptr(), M, C.byref(ws), slabs.data_ptr(), 16, torch.cuda.current_stream().cuda_stream)
        assert ns >= 1, _C.lib().mi355_last_error()
        got = slabs[:ns].sum(0)[:, :N].cpu()
        assert torch.allclose(got, ref32, atol=5e-3, rtol=2e-3), f"{name} slabs ns={ns}: {(got - ref32).abs().max()}"
        img = ops.linear_partial_img(ops.act_image_pack(xd), packed)
        if img is not None:
            assert torch.allclose(img.sum(0)[:, :N].cpu(), ref32, atol=5e-3, rtol=2e-3)
        res = (torch.randn(M, N, generator=torch.Generator().manual_seed(9)) * 2.0).half()
        fused = ops.linear_residual_img(ops.act_image_pack(xd), packed, res.to(DEV))
        if fused is not None:
            assert torch.allclose(fused.cpu().float(), (ref.float() + res.float()).half().float(), **TOL)


def test_linear_partial_slabs_baseline_shapes_vs_oracle():
    """The split-K entry the step driver uses for qkv / o / down (fp32 slabs summed by the consumer): the slab sum must be
    the oracle's fp32 product at the Qwen2-7B shapes for every batch height class."""
    import ctypes as C
    cfg = model.QWEN2_7B
    for name in ("qkv", "o", "down"):
        K, N = _shapes(cfg)[name]
        c_dev = model.synth_linear(K, N, "w4", DEV, torch.Generator(device=DEV).manual_seed(11 + K))
        packed = c_dev.pack()
        W = _dense(model.weights_to({"w": c_dev}, "cpu")["w"])
        x = (torch.randn(64, K, generator=torch.Generator().manual_seed(5)) * 0.5).half()
        ref = x.float() @ W
        ws = ops.weight_struct(packed)
        slabs = torch.zeros(16, 64, packed.N_pad, dtype=torch.float32, device=DEV)
        for M in MS:
            slabs.fill_(float("nan"))
            ns = _C.lib().mi355_linear_partial(x[:M].to(DEV).contiguous().data_ptr(), M, C.byref(ws), slabs.data_ptr(), 16,
                                               torch.cuda.current_stream().cuda_stream)
            assert ns >= 1, _C.lib().mi355_last_error()
            got = slabs.view(-1)[: ns * M * packed.N_pad].view(ns, M, packed.N_pad).sum(0)[:, :N].cpu()
            assert torch.allclose(got, ref[:M], atol=5e-3, rtol=2e-3), f"{name} M={M} ns={ns}: {(got - ref[:M]).abs().max()}"


# ------------------------------------------------------------------ full-width engine step
def _oracle_weights(w):
    return {"embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": _dense(w["lm_head"]),
            "layers": [{"input_norm": L["input_norm"], "post_norm": L["post_norm"], "qkv_bias": L["qkv_bias"],
                        **{k: _dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}


@pytest.mark.parametrize("kind,kv_int8,B,ctx", [("w4", False, 64, 1024), ("w4", True, 64, 4096), ("int8", False, 16, 1024),
                                                ("w4", False, 1, 1024), ("w4", False, 8, 1024)],
                         ids=["w4-b64-ctx1024-kvf16", "w4-b64-ctx4096-kvint8", "w8-b16-ctx1024-kvf16",
                              "w4-b1-fused-small-batch-step", "w4-b8-fused-small-batch-step"])
def test_engine_full_width_step_vs_oracle(kind, kv_int8, B, ctx, parity):
    cfg = model.ModelConfig("qwen2-7b-2l", 2, 3584, 28, 4, 128, 18944, 152064, max_pos=ctx + 16)
    w_dev = model.synth_model(cfg, kind, DEV, seed=21, zeros="centered")   # see synth_linear: realistic zero points
    w = model.weights_to(w_dev, "cpu")
    page, steps = 16, 2
    mb = (ctx + steps + page - 1) // page
    eng = model.DecoderEngine(cfg, w_dev, kv_int8=kv_int8, page=page, num_blocks=B * mb, max_batch=B,
                              max_seq_len=ctx + steps, device=DEV)
    del w_dev
    odec = oracle.OracleDecoder({**cfg.__dict__}, _oracle_weights(w))
    okv = oracle.OracleKV(cfg.num_layers, B, kv_int8)
    g = torch.Generator().manual_seed(5)
    bt = torch.randperm(B * mb, generator=g).reshape(B, mb).to(torch.int32)
    # the same ctx-1 cached tokens on both sides ("allocate KV without prefill", the reference's batch-decode protocol)
    for l in range(cfg.num_layers):
        for b in range(B):
            K = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=g).half()
            V = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=g).half()
            if kv_int8:
                Kq, ks = oracle.quant_kv_int8(K); Vq, vs = oracle.quant_kv_int8(V)
                kvcache.write_tokens(eng.kv[l], eng.kv_scale[l], bt[b], 0, Kq, Vq, ks, vs)
                okv.k[l][b], okv.v[l][b], okv.ks[l][b], okv.vs[l][b] = list(Kq), list(Vq), list(ks), list(vs)
            else:
                kvcache.write_tokens(eng.kv[l], None, bt[b], 0, K, V)
                okv.k[l][b], okv.v[l][b] = list(K), list(V)
    tok = torch.randint(0, cfg.vocab, (B,), generator=g, dtype=torch.int32)
    eng.set_inputs(tok.tolist(), [ctx - 1] * B, bt)
    eng.capture(B)
    for step in range(steps):
        pos = torch.full((B,), ctx - 1 + step, dtype=torch.int32)
        _, ref_logits = odec.forward_tokens(tok, pos, okv, list(range(B)))
        eng.replay(B, 1)
        torch.cuda.synchronize()
        got = eng.logits[:B].cpu()
        assert torch.allclose(got, ref_logits, **TOL), (step, float((got - ref_logits).abs().max()))
        ref_next = oracle.greedy(ref_logits)
        got_next = eng.token_ids[:B].cpu()
        # north_star asks for bit-exact greedy ids: rows whose top-2 margin is below the logits tolerance may legitimately differ,
        # and they are the only ones allowed to; the counts go to the session's parity record (tests/conftest.py)
        rec = parity.step(got_ids=got_next, ref_ids=ref_next, ref_logits=ref_logits, got_logits=got, tol=1e-2, label=f"step {step}")
        assert rec["safe"] >= (B + 1) // 2 and rec["exact"] >= rec["safe"]
        tok = ref_next
        eng.token_ids[:B].copy_(tok)


def test_engine_full_width_step_bf16_vs_oracle(parity):
    """The metric's shapes (Qwen2-7B widths, 2 layers, B = 64, ctx 1024) with bf16 activations and a bf16 KV cache: the staged
    bf16 kernels at their real sizes (gate_up 3584 x 37888 at 64 rows on the 16-wave shape, split-K slabs of qkv / o / down,
    152064-column bf16 lm_head) against the oracle run on bf16 tensors.  Tolerance 1e-2 (see the assert)."""
    BF = torch.bfloat16
    B, ctx = 64, 1024
    cfg = model.ModelConfig("qwen2-7b-2l", 2, 3584, 28, 4, 128, 18944, 152064, max_pos=ctx + 16)
    w_dev = model.synth_model(cfg, "w4", DEV, seed=22, zeros="centered")
    w = model.weights_to(w_dev, "cpu")
    page, steps = 16, 2
    mb = (ctx + steps + page - 1) // page
    eng = model.DecoderEngine(cfg, w_dev, kv_int8=False, page=page, num_blocks=B * mb, max_batch=B, max_seq_len=ctx + steps, device=DEV, dtype=BF)
    del w_dev
    bf = lambda t: None if t is None else t.to(BF)
    dense = lambda c: c.w.to(BF).float() if c.kind == "fp16" else _dense(c)
    ow = {"embedding": bf(w["embedding"]), "final_norm": bf(w["final_norm"]), "lm_head": dense(w["lm_head"]),
          "layers": [{"input_norm": bf(L["input_norm"]), "post_norm": bf(L["post_norm"]), "qkv_bias": bf(L["qkv_bias"]),
                      **{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}
    odec = oracle.OracleDecoder({**cfg.__dict__}, ow)
    okv = oracle.OracleKV(cfg.num_layers, B, False)
    g = torch.Generator().manual_seed(5)
    bt = torch.randperm(B * mb, generator=g).reshape(B, mb).to(torch.int32)
    for l in range(cfg.num_layers):
        for b in range(B):
            K = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=g).to(BF)
            V = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=g).to(BF)
            kvcache.write_tokens(eng.kv[l], None, bt[b], 0, K, V)
            okv.k[l][b], okv.v[l][b] = list(K), list(V)
    tok = torch.randint(0, cfg.vocab, (B,), generator=g, dtype=torch.int32)
    eng.set_inputs(tok.tolist(), [ctx - 1] * B, bt)
    eng.capture(B)
    for step in range(steps):
        pos = torch.full((B,), ctx - 1 + step, dtype=torch.int32)
        _, ref_logits = odec.forward_tokens(tok, pos, okv, list(range(B)))
        eng.replay(B, 1)
        torch.cuda.synchronize()
        got = eng.logits[:B].cpu()
        # north_star's 1e-2, as for fp16: at full width the bf16 step on the image launches (fp16 MFMAs on exact conversions of the bf16
        # activations, fp16 dequant, bf16 stores) measures 5.6e-3 (profiles/r04_parity_greedy_ids.json); the 3e-2 of tests/test_gpu_bf16.py
        # is for its toy widths
        assert torch.allclose(got, ref_logits, atol=1e-2, rtol=1e-2), (step, float((got - ref_logits).abs().max()))
        ref_next = oracle.greedy(ref_logits)
        got_next = eng.token_ids[:B].cpu()
        parity.step(got_ids=got_next, ref_ids=ref_next, ref_logits=ref_logits, got_logits=got, tol=1e-2, label=f"bf16 step {step}")
        tok = ref_next
        eng.token_ids[:B].copy_(tok)


# ------------------------------------------------------------------ large-M GEMM at the real gate_up / down shapes (untested)
PREFILL_CASES = [(m, name) for m in (model.QWEN2_7B, model.LLAMA3_70B) for name in ("gate_up", "down")]


@pytest.mark.parametrize("cfg,name", PREFILL_CASES, ids=[f"{m.name}-{n}" for m, n in PREFILL_CASES])
def test_linear_prefill_baseline_shapes_vs_oracle(cfg, name):
    """M in {128, 200, 1000} rows through mi355_linear_forward at the 7B and 70B gate_up / down shapes: the compute-shaped
    kernel (gemm_prefill.hip; narrow outputs at moderate M stay on 64-row slabs of the decode kernels) against
    oracle.linear.  The oracle multiplies 1000 rows once; smaller M are its leading rows."""
    K, N = _shapes(cfg)[name]
    gen = torch.Generator(device=DEV).manual_seed(K * 3 + N)
    c_dev = model.synth_linear(K, N, "w4", DEV, gen)
    packed = c_dev.pack(gate_up=(name == "gate_up"))
    W = _dense(model.weights_to({"w": c_dev}, "cpu")["w"])
    x = (torch.randn(1000, K, generator=torch.Generator().manual_seed(9)) * 0.5).half()
    ref = oracle.linear(x, W)
    if name == "gate_up":
        ref = oracle.silu_mul(ref)
    del W
    xd = x.to(DEV)
    for M in (128, 200, 1000):
        y = ops.linear(xd[:M].contiguous(), packed, None, epilogue=_C.EPI_SILU_MUL if name == "gate_up" else _C.EPI_NONE)
        torch.cuda.synchronize()
        err = float((y.cpu().float() - ref[:M].float()).abs().max())
        assert torch.allclose(y.cpu().float(), ref[:M].float(), **TOL), f"{cfg.name} {name} M={M}: max err {err}"


def test_prefill_full_width_chunk_vs_oracle():
    """One full-width (Qwen2-7B dims, 2 layers) prefill of three ragged prompts in one 64-row-per-sequence chunk pass: large-M
    GEMMs at the real shapes, rows-mode KV writer, causal multi-row attention, then the logits of each prompt's last token
    against the oracle fed token by token."""
    cfg = model.ModelConfig("qwen2-7b-2l", 2, 3584, 28, 4, 128, 18944, 152064, max_pos=256)
    w_dev = model.synth_model(cfg, "w4", DEV, seed=23, zeros="centered")
    w = model.weights_to(w_dev, "cpu")
    lens, page = [37, 64, 50], 16
    B = len(lens)
    g = torch.Generator().manual_seed(31)
    prompts = [torch.randint(0, cfg.vocab, (n,), generator=g, dtype=torch.int32).tolist() for n in lens]
    eng = model.DecoderEngine(cfg, w_dev, kv_int8=False, page=page, num_blocks=B * 8, max_batch=4, max_seq_len=128, device=DEV)
    del w_dev
    bt = torch.randperm(B * 8, generator=g).reshape(B, 8).to(torch.int32)
    logits = eng.prefill(prompts, bt, chunk=64).cpu()
    odec = oracle.OracleDecoder({**cfg.__dict__}, _oracle_weights(w))
    okv = oracle.OracleKV(cfg.num_layers, B, False)
    ref_last = []
    for b, pr in enumerate(prompts):
        _, lg = odec.forward_tokens(torch.tensor(pr, dtype=torch.int32), torch.arange(len(pr), dtype=torch.int32), okv, [b] * len(pr))
        ref_last.append(lg[-1])
    ref_last = torch.stack(ref_last)
    assert torch.allclose(logits, ref_last, **TOL), float((logits - ref_last).abs().max())
    assert eng.oob_count() == 0


def test_single_layer_uniform_zero_points_vs_oracle():
    """SURVEY 8d's synthetic recipe verbatim (z ~ U{0..15}: weights of mean -scale) at full Qwen2-7B width, ONE layer, B = 64:
    compared at the LAYER OUTPUT (the normed hidden state the engine exposes) instead of the logits -- with these zero points
    the all-ones direction has gain ~40 in the down projection, so a second layer would saturate the softmax, but one layer
    is a well-conditioned check of every kernel on the 8d weights (the end-to-end cases above draw z from {7, 8})."""
    cfg = model.ModelConfig("qwen2-7b-1l", 1, 3584, 28, 4, 128, 18944, 1024, max_pos=64)
    w_dev = model.synth_model(cfg, "w4", DEV, seed=27, zeros="uniform")
    w = model.weights_to(w_dev, "cpu")
    B, page, ctx = 64, 16, 33
    eng = model.DecoderEngine(cfg, w_dev, kv_int8=False, page=page, num_blocks=B * 4, max_batch=B, max_seq_len=64, device=DEV)
    del w_dev
    odec = oracle.OracleDecoder({**cfg.__dict__}, _oracle_weights(w))
    okv = oracle.OracleKV(1, B, False)
    g = torch.Generator().manual_seed(6)
    bt = torch.randperm(B * 4, generator=g).reshape(B, 4).to(torch.int32)
    for b in range(B):
        K = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=g).half(); V = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=g).half()
        kvcache.write_tokens(eng.kv[0], None, bt[b], 0, K, V)
        okv.k[0][b], okv.v[0][b] = list(K), list(V)
    tok = torch.randint(0, cfg.vocab, (B,), generator=g, dtype=torch.int32)
    eng.set_inputs(tok.tolist(), [ctx - 1] * B, bt)
    eng.capture(B); eng.replay(B, 1)
    torch.cuda.synchronize()
    hn_ref, _ = odec.forward_tokens(tok, torch.full((B,), ctx - 1, dtype=torch.int32), okv, list(range(B)))
    hn = eng.hidden[:B].cpu().float()
    # the final RMSNorm brings the residual stream (|h| up to a few hundred here) back to O(1): RMSNorm tolerance of the reference (5e-2 abs)
    # is far too loose for a parity gate, use the linear / attention tolerance
    assert torch.allclose(hn, hn_ref.float(), **TOL), float((hn - hn_ref.float()).abs().max())
