"""bf16 activations (the reference path is fp16 / bf16 throughout: impl/rocm/f16_linear.py:100-112, dtype grid of
modules/base/rocm/test/rocm_norm_test.py): every kernel of the decode step in its bf16 form against the oracle run on bf16 tensors
(the oracle is dtype-generic: fp32 math, rounding to the input dtype at the points the reference's tensors have that dtype).
Tolerance: bf16 keeps 8 significant bits (fp16: 11), so the 1e-2 of the fp16 tests becomes 2e-2 relative + 2e-2 absolute on O(1)
values (one bf16 ulp at 2.0 is 1.6e-2); integer / copy results stay bit-exact."""
import ctypes as C
import math

import pytest
import torch

from oracle import oracle
from rtp_llm_amd import _C, kvcache, model, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
TOL = dict(atol=2e-2, rtol=2e-2)


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def _x(M, K, seed, scale=0.5):
    return (torch.randn(M, K, generator=_gen(seed)) * scale).to(BF)


def _dense(c):
    if c.kind == "int8":
        return oracle.dequant_int8(c.q, c.scales)
    return c.w.to(BF).float() if c.kind == "fp16" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size)


@pytest.mark.parametrize("kind,group", [("w4", 128), ("w4", 64), ("w4", 32), ("int8", 0), ("fp16", 0)])
@pytest.mark.parametrize("M", [1, 7, 16, 33, 64, 83, 200])
@pytest.mark.parametrize("K,N", [(256, 64), (1024, 4608), (3584, 512), (9472, 896)])
def test_linear_bf16(kind, group, M, K, N):
    c = model.synth_linear(K, N, kind, "cpu", _gen(10 + K + N), group or 128)
    x = _x(M, K, M)
    bias = (torch.randn(N, generator=_gen(3)) * 0.1).to(BF) if N % 3 == 0 else None
    w = c.pack(dtype=BF).to(DEV)
    y = ops.linear(x.to(DEV), w, None if bias is None else bias.to(DEV))
    assert y.dtype == BF
    assert torch.allclose(y.cpu().float(), oracle.linear(x, _dense(c), bias).float(), **TOL)
    # fp32 output (lm_head contract): no output rounding left, the accumulator-side dequant is exact up to fp32 summation order
    y32 = ops.linear(x.to(DEV), w, None if bias is None else bias.to(DEV), _C.EPI_OUT_F32)
    ref32 = oracle.linear(x, _dense(c), bias, out_f32=True)
    assert torch.allclose(y32.cpu(), ref32, atol=2e-4, rtol=1e-4), float((y32.cpu() - ref32).abs().max())


def test_linear_bf16_silu_epilogue_and_refusals():
    K, I = 1024, 2432
    c = model.synth_linear(K, 2 * I, "w4", "cpu", _gen(5))
    x = _x(33, K, 6)
    y = ops.linear(x.to(DEV), c.pack(gate_up=True).to(DEV), None, _C.EPI_SILU_MUL)
    ref = oracle.silu_mul(oracle.linear(x, _dense(c)))
    assert y.dtype == BF and torch.allclose(y.cpu().float(), ref.float(), **TOL)
    c16 = model.synth_linear(256, 64, "fp16", "cpu", _gen(8))
    with pytest.raises(_C.Mi355Error):            # a 16-bit weight image has the dtype it was packed with
        ops.linear(_x(4, 256, 1).to(DEV), c16.pack().to(DEV))


@pytest.mark.parametrize("M", [1, 7, 64, 83])
@pytest.mark.parametrize("H", [768, 3584, 8192])
def test_rmsnorm_add_silu_embedding_bf16(M, H):
    x, r = _x(M, H, 1), _x(M, H, 2)
    w = (1 + 0.1 * torch.randn(H, generator=_gen(3))).to(BF)
    y = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6)
    assert y.dtype == BF and torch.allclose(y.cpu().float(), oracle.rmsnorm(x, w, 1e-6).float(), atol=5e-2, rtol=5e-2)
    y2, res = ops.add_rmsnorm(x.to(DEV), r.to(DEV), w.to(DEV), 1e-6)
    assert torch.equal(res.cpu(), x + r)                     # one correctly rounded bf16 add: bit equal
    assert torch.allclose(y2.cpu().float(), oracle.rmsnorm(x + r, w, 1e-6).float(), atol=5e-2, rtol=5e-2)
    gu = _x(M, 2 * H, 5)
    assert torch.allclose(ops.silu_mul(gu.to(DEV)).cpu().float(), oracle.silu_mul(gu).float(), **TOL)
    ids = torch.randint(0, M, (9,), generator=_gen(7), dtype=torch.int32)
    assert torch.equal(ops.embedding(ids.to(DEV), x.to(DEV)).cpu(), x[ids.long()])
    with pytest.raises(_C.Mi355Error):
        ops.rmsnorm(x.to(DEV), w.half().to(DEV), 1e-6)       # mixed dtypes are an error, not a conversion


@pytest.mark.parametrize("nh,nkv,hd,page", [(28, 4, 128, 16), (14, 2, 64, 64), (8, 1, 128, 16)])
def test_rope_kv_write_bf16(nh, nkv, hd, page):
    T, max_blocks, nblk = 6, 8, 64
    cs = oracle.rope_cos_sin(hd, 1e6, max_blocks * page)
    qkv = _x(T, (nh + 2 * nkv) * hd, 11)
    bias = (torch.randn((nh + 2 * nkv) * hd, generator=_gen(12)) * 0.1).to(BF)
    pos = torch.tensor([0, 1, 15, 16, 37, max_blocks * page - 1], dtype=torch.int32)
    bt = torch.randperm(nblk, generator=_gen(2))[: T * max_blocks].reshape(T, max_blocks).to(torch.int32)
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV, dtype=BF)
    q = ops.rope_kv_write(qkv.to(DEV), bias.to(DEV), cs.to(DEV), pos.to(DEV), bt.to(DEV), kv, sc, nh, nkv, hd, page)
    torch.cuda.synchronize()
    qb = (qkv.float() + bias.float()).to(BF)                  # the QKV linear's output tensor (bias inside the linear)
    qh = qb[:, : nh * hd].reshape(T, nh, hd)
    kh = qb[:, nh * hd: (nh + nkv) * hd].reshape(T, nkv, hd)
    vh = qb[:, (nh + nkv) * hd:].reshape(T, nkv, hd)
    q_ref, k_ref = oracle.apply_rope(qh, pos, cs), oracle.apply_rope(kh, pos, cs)
    assert q.dtype == BF and torch.allclose(q.cpu().float(), q_ref.float(), **TOL)
    for t in range(T):
        K, V, _, _ = kvcache.read_tokens(kv, sc, bt[t], int(pos[t]) + 1)
        assert torch.allclose(K[-1].cpu().float(), k_ref[t].float(), **TOL)
        assert torch.equal(V[-1].cpu(), vh[t])               # V is bias add + copy: bit exact
    with pytest.raises(_C.Mi355Error):
        ops.rope_kv_write(qkv.half().to(DEV), None, cs.to(DEV), pos.to(DEV), bt.to(DEV), kv, sc, nh, nkv, hd, page)   # fp16 rows, bf16 cache


def _fill_cache(B, ctx_lens, nkv, hd, page, nblk, seed):
    g = _gen(seed)
    max_blocks = (max(ctx_lens) + page - 1) // page
    bt = torch.randperm(nblk, generator=g)[: B * max_blocks].reshape(B, max_blocks).to(torch.int32)
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV, dtype=BF)
    nat = []
    for b in range(B):
        K = torch.randn(ctx_lens[b], nkv, hd, generator=g).to(BF)
        V = torch.randn(ctx_lens[b], nkv, hd, generator=g).to(BF)
        kvcache.write_tokens(kv, sc, bt[b], 0, K, V)
        nat.append((K, V))
    return kv, sc, bt, nat


@pytest.mark.parametrize("nh,nkv,hd,page", [(28, 4, 128, 16), (32, 8, 128, 16), (14, 2, 64, 64), (8, 1, 128, 64)])
def test_paged_attention_bf16(nh, nkv, hd, page):
    ctx = [1, 7, 8, 15, 16, 17, 31, 33, 127, 128, 129, 500, 1024, 1500]
    B = len(ctx)
    nblk = sum((c + page - 1) // page for c in ctx) + B * ((max(ctx) + page - 1) // page)
    kv, sc, bt, nat = _fill_cache(B, ctx, nkv, hd, page, nblk, 21)
    q = torch.randn(B, nh, hd, generator=_gen(4)).to(BF)
    out = ops.paged_decode_attention(q.to(DEV), kv, sc, bt.to(DEV), torch.tensor(ctx, dtype=torch.int32, device=DEV), nkv, page, max(ctx))
    torch.cuda.synchronize()
    assert out.dtype == BF
    for b in range(B):
        K, V = nat[b]
        ref = oracle.attention_decode(q[b], K, V, 1 / math.sqrt(hd)).reshape(-1)
        assert torch.allclose(out[b].cpu().float(), ref.float(), **TOL), (b, ctx[b], float((out[b].cpu().float() - ref.float()).abs().max()))


def test_paged_attention_rows_causal_bf16():
    nh, nkv, hd, page, q_len, B = 28, 4, 128, 16, 5, 3
    start = [0, 37, 250]
    ctx = [s + q_len for s in start]
    kv, sc, bt, nat = _fill_cache(B, ctx, nkv, hd, page, 128, 31)
    q = torch.randn(B * q_len, nh, hd, generator=_gen(5)).to(BF)
    pos = torch.tensor([s + i for s in start for i in range(q_len)], dtype=torch.int32)
    out = ops.paged_attention_rows(q.to(DEV), kv, sc, bt.to(DEV), pos.to(DEV), nkv, page, q_len, max(ctx))
    for b in range(B):
        K, V = nat[b]
        for i in range(q_len):
            n = start[b] + i + 1
            ref = oracle.attention_decode(q[b * q_len + i], K[:n], V[:n], 1 / math.sqrt(hd)).reshape(-1)
            assert torch.allclose(out[b * q_len + i].cpu().float(), ref.float(), **TOL), (b, i)


def _oracle_weights_bf16(w):
    bf = lambda t: None if t is None else t.to(BF)
    return {"embedding": bf(w["embedding"]), "final_norm": bf(w["final_norm"]), "lm_head": _dense(w["lm_head"]),
            "layers": [{"input_norm": bf(L["input_norm"]), "post_norm": bf(L["post_norm"]), "qkv_bias": bf(L["qkv_bias"]),
                        **{k: _dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}


@pytest.mark.parametrize("kind,B,kv_int8", [("w4", 3, False), ("fp16", 3, False), ("w4", 40, False), ("int8", 3, False), ("w4", 3, True)])
def test_engine_bf16_greedy_decode_matches_oracle(kind, B, kv_int8, parity):
    """The whole decode step in bf16 (DecoderEngine(dtype=torch.bfloat16): bf16 embedding / norms / biases / KV cache, W4 or bf16
    linears through the bf16 MFMA kernels), hipGraph-replayed, against the oracle run on bf16 tensors; prompt fed token by token,
    then greedy generation with the oracle's tokens teacher-forced.  Logits within 3e-2 (three significant bits fewer than fp16
    through 3 layers); greedy ids wherever the oracle's top-2 margin exceeds that.
    Why 3e-2 here and 1e-2 at full width (tests/test_gpu_baseline_shapes.py::test_engine_full_width_step_bf16_vs_oracle, measured 5.6e-3):
    every 16-bit tensor of the step (residual stream, normed rows, q / k / v, the MLP activation) carries a relative rounding error of up to
    2^-9 in bf16 against 2^-12 in fp16, and a logit is a sum over `hidden` such terms -- at this test's hidden = 512 the errors average out
    over 7x fewer terms than at 3584, and the logits themselves are O(5-10); the measured maxima are 1.4e-2 .. 2.9e-2 (profiles/
    r04_parity_greedy_ids.json), i.e. ~2 bf16 ulps of a logit-sized value.  Per-op bf16 checks (this file, TOL = 2e-2) and the full-width step
    carry the tighter bound."""
    cfg = model.ModelConfig("tiny-qwen2", 3, 512, 8, 2, 64, 1024, 2048, max_pos=512)
    w = model.synth_model(cfg, kind, "cpu", seed=3, zeros="centered")
    page, steps = 16, 10
    odec = oracle.OracleDecoder({**cfg.__dict__}, _oracle_weights_bf16(w))
    eng = model.DecoderEngine(cfg, model.weights_to(w, DEV), kv_int8=kv_int8, page=page, num_blocks=4 * B, max_batch=B, max_seq_len=64,
                              device=DEV, dtype=BF)
    assert eng.hidden.dtype == BF and eng.kv[0].dtype == (torch.int8 if kv_int8 else BF)
    bt = torch.randperm(4 * B, generator=_gen(1)).reshape(B, 4).to(torch.int32)

    def kernel_codes(l, b, t):    # INT8 cache: the oracle attends over the codes the kernel wrote (tests/test_gpu_parity.py does the same)
        K, V, ks, vs = kvcache.read_tokens(eng.kv[l], eng.kv_scale[l], bt[b], t + 1)
        return K[t].cpu(), ks[t].cpu(), V[t].cpu(), vs[t].cpu()
    okv = oracle.OracleKV(cfg.num_layers, B, kv_int8, forced=kernel_codes if kv_int8 else None)
    tok = torch.randint(0, cfg.vocab, (B,), generator=_gen(2), dtype=torch.int32)
    eng.set_inputs(tok.tolist(), [0] * B, bt)
    eng.capture(B)
    exact, n, worst = 0, 0, 0.0
    for step in range(steps):
        pos = torch.full((B,), step, dtype=torch.int32)
        eng.replay(B, 1)
        torch.cuda.synchronize()
        _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
        got = eng.logits[:B].cpu()
        assert torch.allclose(got, ref, atol=3e-2, rtol=3e-2), (step, float((got - ref).abs().max()))
        worst = max(worst, float((got - ref).abs().max()))
        ref_next = oracle.greedy(ref)
        got_next = eng.token_ids[:B].cpu()
        exact += parity.step(got_ids=got_next, ref_ids=ref_next, ref_logits=ref, got_logits=got, tol=3e-2, label=f"step {step}")["exact"]; n += B
        assert torch.equal(eng.positions[:B].cpu(), pos + 1)
        tok = ref_next
        eng.token_ids[:B].copy_(tok)
    assert eng.oob_count() == 0
    if kv_int8:   # the scale plane the kernel wrote against the oracle's own scales: one bf16 ulp of a row's amax (2^-8) at most
        assert okv.codes > 0 and okv.max_scale_rel <= 1.6e-2, okv.max_scale_rel
    print(f"bf16 {kind} B={B} kv_int8={kv_int8}: greedy ids {exact}/{n} identical to the oracle's argmax; max |logit error| {worst:.2e}")


def test_engine_bf16_prefill_then_decode_matches_oracle():
    """Ragged prompts through mi355_decoder_prefill in bf16 (64-row slabs of the staged GEMM, rows-mode attention over the bf16 cache),
    then decode steps on the same cache."""
    cfg = model.ModelConfig("tiny-qwen2", 2, 512, 8, 2, 64, 1024, 2048, max_pos=512)
    w = model.synth_model(cfg, "w4", "cpu", seed=5, zeros="centered")
    odec = oracle.OracleDecoder({**cfg.__dict__}, _oracle_weights_bf16(w))
    B, page = 3, 16
    eng = model.DecoderEngine(cfg, model.weights_to(w, DEV), kv_int8=False, page=page, num_blocks=48, max_batch=4, max_seq_len=256,
                              device=DEV, dtype=BF)
    lens = [70, 33, 5]
    prompts = [torch.randint(0, cfg.vocab, (n,), generator=_gen(10 + i), dtype=torch.int32).tolist() for i, n in enumerate(lens)]
    bt = torch.arange(48, dtype=torch.int32).reshape(B, 16)
    logits = eng.prefill(prompts, bt, chunk=32).cpu()
    okv = oracle.OracleKV(cfg.num_layers, B, False)
    for b in range(B):
        _, ref = odec.forward_tokens(torch.tensor(prompts[b], dtype=torch.int32), torch.arange(lens[b], dtype=torch.int32), okv, [b] * lens[b])
        assert torch.allclose(logits[b], ref[-1], atol=3e-2, rtol=3e-2), (b, float((logits[b] - ref[-1]).abs().max()))
    tok = oracle.greedy(logits)
    eng.set_inputs(tok.tolist(), lens, bt)
    for step in range(3):
        pos = torch.tensor([n + step for n in lens], dtype=torch.int32)
        eng.step(B)
        torch.cuda.synchronize()
        _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
        got = eng.logits[:B].cpu()
        assert torch.allclose(got, ref, atol=3e-2, rtol=3e-2), (step, float((got - ref).abs().max()))
        tok = oracle.greedy(ref)
        eng.token_ids[:B].copy_(tok)


def test_engine_dtype_mismatch_is_refused_at_creation():
    """decoder_create checks that every linear and a 16-bit cache carry the step's activation dtype."""
    cfg = model.ModelConfig("tiny-qwen2", 1, 512, 8, 2, 64, 1024, 2048, max_pos=512)
    w = model.synth_model(cfg, "w4", "cpu", seed=6, zeros="centered")
    eng = model.DecoderEngine(cfg, model.weights_to(w, DEV), kv_int8=False, page=16, num_blocks=8, max_batch=2, max_seq_len=64, device=DEV, dtype=BF)
    mc, mw, sb, lw = eng._structs
    lw[0].o.act_dtype = _C.ACT_F16
    assert not _C.lib().mi355_decoder_create(C.byref(mc), lw, C.byref(mw), C.byref(sb)) and b"act_dtype" in _C.lib().mi355_last_error()
    lw[0].o.act_dtype = _C.ACT_BF16
    mc.kv_dtype = _C.KV_FP16
    assert not _C.lib().mi355_decoder_create(C.byref(mc), lw, C.byref(mw), C.byref(sb))


@pytest.mark.parametrize("nh,nkv,hd,page", [(28, 4, 128, 16), (14, 2, 64, 64)])
def test_int8_kv_cache_with_bf16_rows(nh, nkv, hd, page):
    """The INT8 cache under bf16 activations: the writer quantises the bf16-rounded rotated K / V exactly like the oracle
    (integer results bit-exact, K within one code), the attention widens cache bytes to bf16 exactly (v_cvt_f32_ubyte + pack,
    bias 128 carried through both MFMAs) and matches the oracle on the SAME codes."""
    T, max_blocks, nblk = 6, 8, 64
    cs = oracle.rope_cos_sin(hd, 1e6, max_blocks * page)
    qkv = _x(T, (nh + 2 * nkv) * hd, 11)
    pos = torch.tensor([0, 1, 15, 16, 37, max_blocks * page - 1], dtype=torch.int32)
    bt = torch.randperm(nblk, generator=_gen(2))[: T * max_blocks].reshape(T, max_blocks).to(torch.int32)
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, True, DEV)
    q = ops.rope_kv_write(qkv.to(DEV), None, cs.to(DEV), pos.to(DEV), bt.to(DEV), kv, sc, nh, nkv, hd, page)
    assert q.dtype == BF
    kh = qkv[:, nh * hd: (nh + nkv) * hd].reshape(T, nkv, hd)
    vh = qkv[:, (nh + nkv) * hd:].reshape(T, nkv, hd)
    k_ref = oracle.apply_rope(kh, pos, cs)
    for t in range(T):
        K, V, ks, vs = kvcache.read_tokens(kv, sc, bt[t], int(pos[t]) + 1)
        vq, vsc = oracle.quant_kv_int8(vh[t])
        assert torch.equal(V[-1].cpu(), vq) and torch.equal(vs[-1].cpu(), vsc)
        kq, ksc = oracle.quant_kv_int8(k_ref[t])
        assert (K[-1].cpu().int() - kq.int()).abs().max() <= 1 and torch.allclose(ks[-1].cpu(), ksc, rtol=1e-2, atol=0)
    # attention over a cache written by the host with the oracle's codes
    ctx = [1, 8, 17, 33, 129, 500, 1024]
    B = len(ctx)
    g = _gen(21)
    mb = (max(ctx) + page - 1) // page
    nb2 = B * mb
    bt2 = torch.randperm(nb2, generator=g).reshape(B, mb).to(torch.int32)
    kv2, sc2 = kvcache.alloc_layer_cache(nb2, nkv, page, hd, True, DEV)
    nat = []
    for b in range(B):
        K = torch.randn(ctx[b], nkv, hd, generator=g).to(BF); V = torch.randn(ctx[b], nkv, hd, generator=g).to(BF)
        Kq, ks = oracle.quant_kv_int8(K); Vq, vs = oracle.quant_kv_int8(V)
        kvcache.write_tokens(kv2, sc2, bt2[b], 0, Kq, Vq, ks, vs)
        nat.append((Kq, Vq, ks, vs))
    qq = torch.randn(B, nh, hd, generator=_gen(4)).to(BF)
    out = ops.paged_decode_attention(qq.to(DEV), kv2, sc2, bt2.to(DEV), torch.tensor(ctx, dtype=torch.int32, device=DEV), nkv, page, max(ctx))
    assert out.dtype == BF
    for b in range(B):
        Kq, Vq, ks, vs = nat[b]
        ref = oracle.attention_decode(qq[b], Kq, Vq, 1 / math.sqrt(hd), ks, vs).reshape(-1)
        assert torch.allclose(out[b].cpu().float(), ref.float(), **TOL), (b, ctx[b], float((out[b].cpu().float() - ref.float()).abs().max()))


@pytest.mark.parametrize("M", [3, 17, 64])
def test_image_path_layer_with_bf16_tensors_vs_oracle(M):
    """The launches on activation images (gemm_fullk64 / gemm_wide image entry / gemm_splitk64) inside a bf16 step: the images hold
    fp16 conversions of the bf16 values (exact), the GEMMs run fp16 MFMAs on the fp16 dequant, the epilogues round and store bf16
    (q, KV cache, residual stream).  One attention-less layer chain against the oracle on bf16 tensors:
      RMSNorm -> [image] QKV + bias + RoPE + KV write;  attn (random) -> [image] O + residual (+ deferred norm operands)
      -> [image] gate_up + SiLU -> [image] down slabs -> fold (residual + RMSNorm) -> [image]."""
    cfg = model.QWEN2_7B
    H, I, nh, nkv, hd, page = cfg.hidden, cfg.inter, cfg.nh, cfg.nkv, cfg.hd, 16
    g = _gen(M)
    mk = lambda K, N, seed, **kw: model.synth_linear(K, N, "w4", "cpu", _gen(seed), zeros="centered")
    cq, co, cg, cd = mk(H, (nh + 2 * nkv) * hd, 1), mk(H, H, 2), mk(H, 2 * I, 3), mk(I, H, 4)
    wq, wo, wg, wd = cq.pack(dtype=BF).to(DEV), co.pack(dtype=BF).to(DEV), cg.pack(gate_up=True, dtype=BF).to(DEV), cd.pack(dtype=BF).to(DEV)
    Wq, Wo, Wg, Wd = _dense(cq), _dense(co), _dense(cg), _dense(cd)
    h0 = (torch.randn(M, H, generator=g) * 2.0).to(BF)
    g_in = (1.0 + 0.2 * torch.randn(H, generator=g)).to(BF); g_post = (1.0 + 0.2 * torch.randn(H, generator=g)).to(BF)
    bias = (torch.randn((nh + 2 * nkv) * hd, generator=g) * 0.1).to(BF)
    eps = 1e-6
    # ---- QKV on the image the norm writes
    xn_img, _ = ops.add_rmsnorm_img(h0.to(DEV), None, g_in.to(DEV), eps)
    assert xn_img.data.dtype == torch.float16
    xn_ref = oracle.rmsnorm(h0, g_in, eps)
    # the bf16 result x 2^-8 as fp16 (csrc/common.h img_val): exact down to |x| = 2^-6, an absolute spacing of 2^-16 below
    got_xn = xn_img.unpack().cpu()
    assert got_xn.dtype == BF and torch.allclose(got_xn.float(), xn_ref.float(), atol=2.0 ** -17, rtol=0)
    assert torch.equal(got_xn[xn_ref.abs() >= 2.0 ** -6], xn_ref[xn_ref.abs() >= 2.0 ** -6])
    max_blocks, nblk = 8, 1024
    c2 = model.ModelConfig("t", 1, H, nh, nkv, hd, 64, 128, max_pos=max_blocks * page)
    cs = oracle.rope_cos_sin(hd, c2.rope_theta, c2.max_pos)
    pos = torch.randint(0, max_blocks * page, (M,), generator=g).to(torch.int32)
    bt = torch.randperm(nblk, generator=g)[: M * max_blocks].reshape(M, max_blocks).to(torch.int32)
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV, dtype=BF)
    q = ops.qkv_rope_kv_write_img(xn_img, wq, bias.to(DEV), cs.to(DEV), pos.to(DEV), bt.to(DEV), kv, sc, nh, nkv, hd, page)
    assert q is not None and q.dtype == BF
    qkv = oracle.linear(xn_ref, Wq, bias)
    q_ref = oracle.apply_rope(qkv[:, : nh * hd].reshape(M, nh, hd), pos, cs)
    k_ref = oracle.apply_rope(qkv[:, nh * hd: (nh + nkv) * hd].reshape(M, nkv, hd), pos, cs)
    assert torch.allclose(q.cpu().float(), q_ref.float(), **TOL)
    for t in range(M):
        K, V, _, _ = kvcache.read_tokens(kv, sc, bt[t], int(pos[t]) + 1)
        assert torch.allclose(K[-1].cpu().float(), k_ref[t].float(), **TOL)
        assert torch.allclose(V[-1].cpu().float(), qkv[t, (nh + nkv) * hd:].reshape(nkv, hd).float(), **TOL)
    # ---- O + residual with the deferred norm, gate_up + SiLU as an image, down slabs, fold
    attn = (torch.randn(M, nh * hd, generator=g) * 0.5).to(BF)
    r = ops.linear_residual_prenorm_img(ops.act_image_pack(attn.to(DEV)), wo, h0.to(DEV), g_post.to(DEV))
    assert r is not None
    h1, xg, ssq, e = r
    h1_ref = (oracle.linear(attn, Wo, None).float() + h0.float()).to(BF)
    assert h1.dtype == BF and torch.allclose(h1.cpu().float(), h1_ref.float(), **TOL)
    act_img = ops.linear_deferred_norm_img(xg, (ssq, eps, e), wg, None, _C.EPI_SILU_MUL | _C.EPI_OUT_IMAGE, act=BF)
    assert isinstance(act_img, ops.ActImage)
    act_ref = oracle.silu_mul(oracle.linear(oracle.rmsnorm(h1.cpu(), g_post, eps), Wg, None))
    act = act_img.unpack().cpu()
    assert act.dtype == BF                                                                # the image stands for a bf16 tensor
    assert torch.allclose(act.float(), act_ref.float(), **TOL), float((act.float() - act_ref.float()).abs().max())
    slabs = ops.linear_partial_img(act_img, wd)
    assert slabs is not None
    down_ref = oracle.linear(act.to(BF), Wd, None)
    assert torch.allclose(slabs.sum(0)[:, :H].cpu(), down_ref.float(), atol=4e-2, rtol=2e-2)


def test_bf16_images_carry_activations_beyond_the_fp16_range():
    """bf16 reaches 3.4e38, an activation image stores fp16: the image of a bf16 tensor holds x 2^-8 (csrc/common.h img_val) and the GEMMs
    that read it multiply their accumulators by 2^8, so the "massive activations" of bf16 checkpoints -- the SiLU * up product above
    all -- pass through the image launches: values up to 1.6e7 instead of inf beyond 65504.  Checked on the two places
    they occur: down_proj reading such an image, and gate_up's SiLU epilogue writing one."""
    cfg = model.QWEN2_7B
    H, I, M = cfg.hidden, cfg.inter, 17
    g = _gen(99)
    cd = model.synth_linear(I, H, "w4", "cpu", _gen(5), zeros="centered")
    wd, Wd = cd.pack(dtype=BF).to(DEV), _dense(cd)
    x = torch.randn(M, I, generator=g)
    big = torch.rand(M, I, generator=g) < 1e-3                      # a few elements per row far beyond the fp16 range
    x = torch.where(big, x * 4e5, x).to(BF)
    assert float(x.float().abs().max()) > 65504 * 4
    img = ops.act_image_pack(x.to(DEV))
    assert torch.isfinite(img.data.float()).all()
    back = img.unpack().cpu()
    keep = x.abs() >= 2.0 ** -6
    assert torch.equal(back[keep], x[keep]) and torch.allclose(back.float(), x.float(), atol=2.0 ** -17, rtol=0)
    slabs = ops.linear_partial_img(img, wd)
    assert slabs is not None
    got, ref = slabs.sum(0)[:, :H].cpu(), oracle.linear(x, Wd, None, out_f32=True)
    assert torch.isfinite(got).all() and torch.allclose(got, ref, atol=2e-3 * float(ref.abs().max()), rtol=1e-2), float((got - ref).abs().max())
    # gate_up with scales x 700: gate and up are O(300) (xavier weights give O(0.4) on normalised rows), their SiLU product reaches 1e5-1e6
    cg = model.synth_linear(H, 2 * I, "w4", "cpu", _gen(6), zeros="centered")
    cg.scales = (cg.scales.float() * 700).half()
    wg, Wg = cg.pack(gate_up=True, dtype=BF).to(DEV), _dense(cg)
    co = model.synth_linear(H, H, "w4", "cpu", _gen(7), zeros="centered")
    wo = co.pack(dtype=BF).to(DEV)
    h0 = (torch.randn(M, H, generator=g) * 2.0).to(BF)
    g_post = (1.0 + 0.2 * torch.randn(H, generator=g)).to(BF)
    attn = (torch.randn(M, H, generator=g) * 0.5).to(BF)
    h1, xg, ssq, e = ops.linear_residual_prenorm_img(ops.act_image_pack(attn.to(DEV)), wo, h0.to(DEV), g_post.to(DEV))
    act_img = ops.linear_deferred_norm_img(xg, (ssq, 1e-6, e), wg, None, _C.EPI_SILU_MUL | _C.EPI_OUT_IMAGE, act=BF)
    act_ref = oracle.silu_mul(oracle.linear(oracle.rmsnorm(h1.cpu(), g_post, 1e-6), Wg, None))
    assert float(act_ref.float().abs().max()) > 2 * 65504, float(act_ref.float().abs().max())
    assert torch.isfinite(act_img.data.float()).all()
    act = act_img.unpack().cpu().float()
    # what is checked here is the RANGE (finite, right magnitude everywhere), not the last bits: the product of two O(300) factors, each
    # rounded to bf16 on both sides from GEMM sums that differ in the last fp16-operand bits (a weight scale of 1.1 makes those 0.3
    # absolute), rounded to bf16 again: 5e-2 relative + 2e-3 of the largest element
    ref = act_ref.float()
    excess = (act - ref).abs() - (2e-3 * float(ref.abs().max()) + 5e-2 * ref.abs())
    i = int(excess.argmax())
    assert float(excess.max()) <= 0, (f"worst element: got {float(act.reshape(-1)[i])} want {float(ref.reshape(-1)[i])}, max |ref| {float(ref.abs().max())}, "
                                       f"{int((excess > 0).sum())} of {excess.numel()} outside")
