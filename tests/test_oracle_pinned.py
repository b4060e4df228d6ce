"""Pins oracle/oracle.py to the reference for the floating-point layer math.

tests/golden/ref_layers.npz holds inputs and outputs of the reference's OWN torch reference implementations,
executed unmodified from /root/reference by oracle/gen_golden.py (the functions its ROCm unit tests compare the native
kernels against):
    RMSNormTorch                        models_py/modules/base/common/norm.py:83-92
    _torch_reference (NeoX RoPE)        models_py/modules/factory/attention/rocm_impl/test/test_fused_qkv_transpose_v3.py:246-307
    run_native / ref_masked_attention   models_py/modules/base/rocm/test/rocm_fmha_test.py:262-372
    DenseMLP                            models_py/modules/hybrid/test/dense_mlp_ref.py:11-34
The oracle must reproduce them: bit for bit where both sides perform the same fp32 operations, and to within fp32
re-association noise (far below one fp16 ulp of the outputs) where the contraction order differs.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import oracle


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_layers.npz"))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_rmsnorm_reproduces_reference_rmsnormtorch(ref):
    for i in range(int(ref["norm_count"])):
        x, w, y = _t(ref[f"norm{i}_x"]), _t(ref[f"norm{i}_w"]), _t(ref[f"norm{i}_y"])
        assert torch.equal(oracle.rmsnorm(x, w, 1e-6), y), f"case {i}"


def test_rope_reproduces_reference_torch_reference(ref):
    for i in range(int(ref["rope_count"])):
        nh, nkv, hd, base = ref[f"rope{i}_cfg"]
        nh, nkv, hd = int(nh), int(nkv), int(hd)
        qkv, lens = _t(ref[f"rope{i}_qkv"]), ref[f"rope{i}_lens"].tolist()
        pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens])
        T = qkv.shape[0]
        cs = oracle.rope_cos_sin(hd, float(base), max(lens))
        q = oracle.apply_rope(qkv[:, : nh * hd].reshape(T, nh, hd), pos, cs)
        k = oracle.apply_rope(qkv[:, nh * hd: (nh + nkv) * hd].reshape(T, nkv, hd), pos, cs)
        v = qkv[:, (nh + nkv) * hd:].reshape(T, nkv, hd)
        q_ref, k_ref = _t(ref[f"rope{i}_q"]), _t(ref[f"rope{i}_k"])
        assert torch.equal(v, _t(ref[f"rope{i}_v"]))
        # the reference test builds inv_freq as base ** (-2i/dim), the engine's table (RopeCache.cc:16-41) as
        # 1 / base ** (2i/dim): the angles differ in the last fp32 bit, the rotated fp16 values by at most one ulp
        for got, want in ((q, q_ref), (k, k_ref)):
            d = (got.float() - want.float()).abs()
            ulp = torch.exp2(torch.floor(torch.log2(torch.clamp(torch.maximum(want.float().abs(), got.float().abs()), min=2.0 ** -14))) - 10)
            # (near-cancelling rotations sit in a lower binade than their operands: absolute floor of 1e-6 there)
            assert bool((d <= torch.clamp(ulp, min=1e-6)).all()), f"case {i}: max {d.max()}"
            assert (d > 0).float().mean() < 1e-3, f"case {i}: {(d > 0).float().mean():.4f} of the elements differ"


def test_attention_reproduces_reference_run_native(ref):
    for i in range(int(ref["attn_count"])):
        nh, nkv, hd, block, int8 = (int(v) for v in ref[f"attn{i}_cfg"])
        ctx, bt, q = ref[f"attn{i}_ctx"].tolist(), _t(ref[f"attn{i}_bt"]), _t(ref[f"attn{i}_q"])
        K, V, out = _t(ref[f"attn{i}_k"]), _t(ref[f"attn{i}_v"]), _t(ref[f"attn{i}_out"])
        for b, n in enumerate(ctx):
            idx = torch.tensor([int(bt[b, j // block]) * block + j % block for j in range(n)])
            ks = vs = None
            if int8:
                ks, vs = _t(ref[f"attn{i}_ks"])[:, idx].t(), _t(ref[f"attn{i}_vs"])[:, idx].t()   # [ctx, nkv]
            got = oracle.attention_decode(q[b], K[idx], V[idx], 1.0 / math.sqrt(hd), ks, vs)
            # same fp32 formula; einsum contraction order may differ: allow fp32 noise, i.e. at most one fp16 ulp on
            # a handful of outputs
            d = (got.float() - out[b].float()).abs()
            assert d.max() <= 2.0 ** -10 * max(1.0, float(out[b].float().abs().max())), (i, b, float(d.max()))
            assert (d > 0).float().mean() < 0.01, (i, b)


def test_int8_kv_quantiser_matches_the_fixture_convention(ref):
    """attn3/attn4 were quantised by the generator with scale = amax/127, rne, saturate — the oracle's quant_kv_int8
    must be that same function (it is what the HIP writer is checked against)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(50, 3, 128, generator=g).half()
    q, s = oracle.quant_kv_int8(x)
    amax = x.float().abs().amax(-1)
    assert torch.equal(s, amax / 127.0)
    assert torch.equal(q, torch.clamp(torch.round(x.float() / s.unsqueeze(-1)), -128, 127).to(torch.int8))
    assert int(q.abs().max()) == 127


def test_mlp_reproduces_reference_dense_mlp(ref):
    """The reference MLP keeps gate / up as separate fp16 linears (torch fp16 matmul on CPU accumulates in fp32 and
    rounds once, as the oracle's linear does) and multiplies silu(gate) * up in fp16; the oracle's fused form rounds
    the product once from fp32.  They agree to one fp16 rounding of the activation, far inside the 1e-2 gate."""
    x, gate, up, down, y = (_t(ref[k]) for k in ("mlp_x", "mlp_gate", "mlp_up", "mlp_down", "mlp_y"))
    gu = oracle.linear(x, torch.cat([gate, up], dim=1).float())
    got = oracle.linear(oracle.silu_mul(gu), down.float())
    assert torch.allclose(got.float(), y.float(), atol=2e-3, rtol=2e-3), float((got.float() - y.float()).abs().max())


# ------------------------------------------------------------------ the same pins in bf16 (the reference's second activation dtype)
@pytest.fixture(scope="module")
def ref_bf16(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_layers_bf16.npz"))


def _bf(a):
    return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16)


def test_bf16_oracle_reproduces_the_reference_torch_references(ref_bf16):
    """tests/golden/ref_layers_bf16.npz: RMSNormTorch, _torch_reference (RoPE), run_native (paged attention) and DenseMLP of the
    reference executed on bf16 tensors (oracle/gen_golden.py:gen_ref_layers_bf16; arrays stored as int16 bit patterns).  The
    oracle run on the same bf16 tensors must reproduce them: this is what the bf16 HIP kernels are compared against."""
    r = ref_bf16
    for i in range(int(r["norm_count"])):
        x, w, y = _bf(r[f"norm{i}_x"]), _bf(r[f"norm{i}_w"]), _bf(r[f"norm{i}_y"])
        assert torch.equal(oracle.rmsnorm(x, w, 1e-6), y), f"norm case {i}"
    for i in range(int(r["rope_count"])):
        nh, nkv, hd, base = r[f"rope{i}_cfg"]
        nh, nkv, hd = int(nh), int(nkv), int(hd)
        qkv, lens = _bf(r[f"rope{i}_qkv"]), r[f"rope{i}_lens"].tolist()
        pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens])
        T = qkv.shape[0]
        cs = oracle.rope_cos_sin(hd, float(base), max(lens))
        q = oracle.apply_rope(qkv[:, : nh * hd].reshape(T, nh, hd), pos, cs)
        k = oracle.apply_rope(qkv[:, nh * hd: (nh + nkv) * hd].reshape(T, nkv, hd), pos, cs)
        assert torch.equal(qkv[:, (nh + nkv) * hd:].reshape(T, nkv, hd), _bf(r[f"rope{i}_v"]))
        for got, want in ((q, _bf(r[f"rope{i}_q"])), (k, _bf(r[f"rope{i}_k"]))):
            d = (got.float() - want.float()).abs()
            ulp = torch.exp2(torch.floor(torch.log2(torch.clamp(torch.maximum(want.float().abs(), got.float().abs()), min=2.0 ** -14))) - 7)
            assert bool((d <= torch.clamp(ulp, min=1e-6)).all()), f"rope case {i}: max {d.max()}"      # last-bit angle differences: <= 1 bf16 ulp
            assert (d > 0).float().mean() < 1e-3, f"rope case {i}: {(d > 0).float().mean():.4f} of the elements differ"
    for i in range(int(r["attn_count"])):
        nh, nkv, hd, block = (int(v) for v in r[f"attn{i}_cfg"])
        ctx, bt, q = r[f"attn{i}_ctx"].tolist(), _t(r[f"attn{i}_bt"]), _bf(r[f"attn{i}_q"])
        K, V, out = _bf(r[f"attn{i}_k"]), _bf(r[f"attn{i}_v"]), _bf(r[f"attn{i}_out"])
        for b, n in enumerate(ctx):
            idx = torch.tensor([int(bt[b, j // block]) * block + j % block for j in range(n)])
            got = oracle.attention_decode(q[b], K[idx], V[idx], 1.0 / math.sqrt(hd))
            d = (got.float() - out[b].float()).abs()
            assert d.max() <= 2.0 ** -7 * max(1.0, float(out[b].float().abs().max())), (i, b, float(d.max()))   # <= one bf16 ulp
            assert (d > 0).float().mean() < 0.01, (i, b)
    x, gate, up, down, y = (_bf(r[k]) for k in ("mlp_x", "mlp_gate", "mlp_up", "mlp_down", "mlp_y"))
    gu = oracle.linear(x, torch.cat([gate, up], dim=1).float())
    got = oracle.linear(oracle.silu_mul(gu), down.float())
    assert torch.allclose(got.float(), y.float(), atol=1.6e-2, rtol=1.6e-2), float((got.float() - y.float()).abs().max())
