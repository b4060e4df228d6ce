"""Generate golden vectors by running the REFERENCE's own load-time quantisation code.

TEST INFRASTRUCTURE ONLY.  Runs in the authoring container only (needs /root/reference);
the committed outputs (tests/golden/quant_*.npz) are what travels.

The reference's ``rtp_llm/device/device_impl.py`` is pure Python over torch once its
imports of the compiled op libraries are stubbed (verified importable with
``sys.modules`` stubs).  We execute, unmodified:
  * RocmImpl.preprocess_groupwise_weight_params (GPTQ and AWQ, 4-bit)  device_impl.py:797-868
  * GpuImpl.apply_int8 / symmetric_quantize_last_axis_of_batched_matrix  device_impl.py:183-222
and record inputs + outputs.  We also execute the reference's own torch reference implementations (the functions
its ROCm unit tests compare the native kernels against) and record their inputs + outputs in
tests/golden/ref_layers.npz, so that oracle/oracle.py is pinned to the reference for the floating-point layer math:
  * RMSNormTorch.forward                 models_py/modules/base/common/norm.py:83-92
  * ref_masked_attention / run_native    models_py/modules/base/rocm/test/rocm_fmha_test.py:262-372
  * _torch_reference (NeoX RoPE, QKV split)  models_py/modules/factory/attention/rocm_impl/test/test_fused_qkv_transpose_v3.py:246-307
  * DenseMLP (SiLU-gate MLP)             models_py/modules/hybrid/test/dense_mlp_ref.py:11-34
From the ROCm-packed int4 output we also recover the canonical
codes (undoing the CK nibble permutation and column-major packing, device_impl.py:729-771)
so tests can pin rtp_llm_amd.quant.unpack_gptq/unpack_awq bit-exactly.

    python oracle/gen_golden.py            # writes tests/golden/quant_{gptq,awq,int8}.npz and ref_layers.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class _Any:
    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


class _Stub(types.ModuleType):
    """Stand-in for the compiled / unrelated rtp_llm modules: any attribute resolves."""
    __path__ = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Any()


def _install_stubs():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; golden vectors are committed under tests/golden/")
    for name in ["rtp_llm", "rtp_llm.ops", "rtp_llm.ops.compute_ops", "rtp_llm.config", "rtp_llm.config.py_config_modules",
                 "rtp_llm.utils", "rtp_llm.utils.model_weight", "rtp_llm.utils.swizzle_utils", "rtp_llm.device",
                 "rtp_llm.device.device_type"]:
        sys.modules[name] = _Stub(name)
    if "psutil" not in sys.modules:
        try:
            import psutil  # noqa: F401
        except ImportError:
            sys.modules["psutil"] = _Stub("psutil")
    # rocm_fmha_test.py compares cache dtypes against aiter.dtypes.{fp8,i8}; dense_mlp_ref.py against ActivationType.Swiglu
    aiter = _Stub("aiter")
    aiter.dtypes = types.SimpleNamespace(fp8=torch.float8_e4m3fn, i8=torch.int8, fp16=torch.float16, bf16=torch.bfloat16, fp32=torch.float32)
    sys.modules["aiter"] = aiter
    sys.modules["rtp_llm.ops"].ActivationType = types.SimpleNamespace(Swiglu="Swiglu")


def _load(modname, rel):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def _import_reference_device_impl():
    _install_stubs()
    _load("rtp_llm.device.device_base", "rtp_llm/device/device_base.py")
    return _load("rtp_llm.device.device_impl", "rtp_llm/device/device_impl.py")


def _make_rocm_impl(mod):
    class Impl(mod.RocmImpl):
        def __init__(self):  # skip DeviceBase/rocml init: only the pure-torch math is exercised
            self.rocml = None

        @property
        def specify_gpu_arch(self):
            return ""
    return Impl()


def _rand_gptq(K, N, g, seed):
    gen = torch.Generator().manual_seed(seed)
    qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int64, generator=gen).to(torch.int32)
    qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // g, N // 8), dtype=torch.int64, generator=gen).to(torch.int32)
    scales = (torch.rand(K // g, N, generator=gen) * 0.02 + 0.005).half()
    return qweight, qzeros, scales


def _rand_awq(K, N, g, seed):
    gen = torch.Generator().manual_seed(seed)
    qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int64, generator=gen).to(torch.int32)
    qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // g, N // 8), dtype=torch.int64, generator=gen).to(torch.int32)
    scales = (torch.rand(K // g, N, generator=gen) * 0.02 + 0.005).half()
    return qweight, qzeros, scales


def gen_ref_layers():
    """Run the reference's own torch reference implementations on seeded inputs -> tests/golden/ref_layers.npz."""
    _install_stubs()
    MP = "rtp_llm/models_py/"
    norm = _load("ref_norm", MP + "modules/base/common/norm.py")
    # rocm_fmha_test.py queries the device at import time (gcnArchName, to skip Navi): answer for the absent GPU
    real_props = torch.cuda.get_device_properties
    torch.cuda.get_device_properties = lambda *a, **k: types.SimpleNamespace(gcnArchName="gfx950")
    try:
        fmha = _load("ref_fmha_test", MP + "modules/base/rocm/test/rocm_fmha_test.py")
    finally:
        torch.cuda.get_device_properties = real_props
    rope = _load("ref_rope_test", MP + "modules/factory/attention/rocm_impl/test/test_fused_qkv_transpose_v3.py")
    mlp = _load("ref_dense_mlp", MP + "modules/hybrid/test/dense_mlp_ref.py")
    out = {}
    g = torch.Generator().manual_seed(1234)
    rn = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s)

    # ---- RMSNormTorch (norm.py:83-92): token counts / hidden sizes from rocm_norm_test.py's grid + the model widths
    for i, (T, H) in enumerate([(7, 768), (23, 896), (5, 3584), (3, 8192), (2, 8199)]):
        x, w = rn(T, H).half(), rn(H).half()
        y = norm.RMSNormTorch(w, 1e-6)(x)
        out[f"norm{i}_x"], out[f"norm{i}_w"], out[f"norm{i}_y"] = x.numpy(), w.numpy(), y.numpy()
    out["norm_count"] = np.array(5)

    # ---- _torch_reference (test_fused_qkv_transpose_v3.py:246-307): NeoX RoPE on packed QKV, positions 0..len-1 per request
    for i, (nh, nkv, hd, base, lens) in enumerate([(28, 4, 128, 1e6, [5, 11]), (14, 2, 64, 1e6, [19]), (8, 1, 128, 5e5, [3, 1, 9])]):
        T = sum(lens)
        qkv = rn(T, (nh + 2 * nkv) * hd).half()
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
        Q, K, V = rope._torch_reference(qkv, nh, nkv, hd, hd, base, 1.0, cu)
        out[f"rope{i}_qkv"], out[f"rope{i}_cfg"], out[f"rope{i}_lens"] = qkv.numpy(), np.array([nh, nkv, hd, base]), np.array(lens)
        out[f"rope{i}_q"], out[f"rope{i}_k"], out[f"rope{i}_v"] = Q.numpy(), K.numpy(), V.numpy()
    out["rope_count"] = np.array(3)

    # ---- run_native / ref_masked_attention (rocm_fmha_test.py:262-372): paged decode attention, caches in the
    # reference test's layouts K [blocks, nkv, hd/x, block, x], V [blocks, nkv, hd, block]  (x = 16 / elt bytes)
    def attn_case(i, nh, nkv, hd, block, ctx, int8):
        B, x = len(ctx), (16 if int8 else 8)
        mb = (max(ctx) + block - 1) // block
        nblk = B * mb
        bt = torch.randperm(nblk, generator=g).reshape(B, mb).to(torch.int32)
        q = rn(B, nh, hd).half()
        Kn, Vn = rn(nblk * block, nkv, hd).half(), rn(nblk * block, nkv, hd).half()     # natural [slot, nkv, hd]
        if int8:   # per (token, kv head) scale planes [nkv, slots]; the convention under test is the oracle's quant_kv_int8
            ka, va = Kn.float().abs().amax(-1), Vn.float().abs().amax(-1)
            ks, vs = (ka / 127.0), (va / 127.0)
            Kc = torch.clamp(torch.round(Kn.float() / ks.unsqueeze(-1)), -128, 127).to(torch.int8)
            Vc = torch.clamp(torch.round(Vn.float() / vs.unsqueeze(-1)), -128, 127).to(torch.int8)
            k_scale, v_scale = ks.t().contiguous(), vs.t().contiguous()
        else:
            Kc, Vc = Kn, Vn
            k_scale = v_scale = torch.tensor([1.0])
        k_cache = Kc.view(nblk, block, nkv, hd // x, x).permute(0, 2, 3, 1, 4).contiguous()
        v_cache = Vc.view(nblk, block, nkv, hd).permute(0, 2, 3, 1).contiguous()
        o = fmha.run_native(q, k_cache, v_cache, bt, torch.tensor(ctx), max(ctx), "auto", nkv, 1.0 / hd ** 0.5, None,
                            k_scale, v_scale, nh // nkv, torch.float16)
        out[f"attn{i}_cfg"] = np.array([nh, nkv, hd, block, int(int8)])
        out[f"attn{i}_ctx"], out[f"attn{i}_bt"], out[f"attn{i}_q"] = np.array(ctx), bt.numpy(), q.numpy()
        out[f"attn{i}_k"], out[f"attn{i}_v"], out[f"attn{i}_out"] = Kc.numpy(), Vc.numpy(), o.numpy()
        if int8:
            out[f"attn{i}_ks"], out[f"attn{i}_vs"] = k_scale.numpy(), v_scale.numpy()
    attn_case(0, 28, 4, 128, 16, [1, 17, 70], False)      # Qwen2-7B heads
    attn_case(1, 14, 2, 64, 16, [33, 64], False)                # Qwen2-0.5B heads
    attn_case(2, 8, 1, 128, 16, [5, 100], False)                # Llama-3-70B per-rank heads at tp = 8
    # 8-bit KV with per-token scales: ref_masked_attention views the head axis as [group, kv head] when it applies the
    # scales while run_native expands K/V kv-head-major (repeat_interleave), so the two agree with each other only for
    # one kv head (any group size) or group size 1 -- those are the cases that pin the oracle's scale handling
    attn_case(3, 8, 1, 128, 16, [40, 100], True)
    attn_case(4, 4, 4, 64, 16, [70], True)
    out["attn_count"] = np.array(5)

    # ---- DenseMLP torch reference (dense_mlp_ref.py:11-34): weights [in, out]
    H, I, T = 256, 640, 9
    x, gate, up, down = rn(T, H, s=0.5).half(), rn(H, I, s=0.06).half(), rn(H, I, s=0.06).half(), rn(I, H, s=0.04).half()
    y = mlp.DenseMLP(gate, up, down, sys.modules["rtp_llm.ops"].ActivationType.Swiglu)(x)
    out.update(mlp_x=x.numpy(), mlp_gate=gate.numpy(), mlp_up=up.numpy(), mlp_down=down.numpy(), mlp_y=y.numpy())
    np.savez_compressed(os.path.join(OUT, "ref_layers.npz"), **out)


def gen_ref_layers_bf16():
    """The same reference implementations on bf16 tensors (the reference's second activation dtype) ->
    tests/golden/ref_layers_bf16.npz; bf16 arrays are stored as their int16 bit patterns (numpy has no bf16)."""
    _install_stubs()
    MP = "rtp_llm/models_py/"
    norm = _load("ref_norm", MP + "modules/base/common/norm.py")
    real_props = torch.cuda.get_device_properties
    torch.cuda.get_device_properties = lambda *a, **k: types.SimpleNamespace(gcnArchName="gfx950")
    try:
        fmha = _load("ref_fmha_test", MP + "modules/base/rocm/test/rocm_fmha_test.py")
    finally:
        torch.cuda.get_device_properties = real_props
    rope = _load("ref_rope_test", MP + "modules/factory/attention/rocm_impl/test/test_fused_qkv_transpose_v3.py")
    mlp = _load("ref_dense_mlp", MP + "modules/hybrid/test/dense_mlp_ref.py")
    BF = torch.bfloat16
    bits = lambda t: t.contiguous().view(torch.int16).numpy()
    out = {}
    g = torch.Generator().manual_seed(4321)
    rn = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s)
    for i, (T, H) in enumerate([(7, 896), (5, 3584), (2, 8199)]):
        x, w = rn(T, H).to(BF), rn(H).to(BF)
        y = norm.RMSNormTorch(w, 1e-6)(x)
        assert y.dtype == BF
        out[f"norm{i}_x"], out[f"norm{i}_w"], out[f"norm{i}_y"] = bits(x), bits(w), bits(y)
    out["norm_count"] = np.array(3)
    for i, (nh, nkv, hd, base, lens) in enumerate([(28, 4, 128, 1e6, [5, 11]), (8, 1, 128, 5e5, [3, 1, 9])]):
        T = sum(lens)
        qkv = rn(T, (nh + 2 * nkv) * hd).to(BF)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
        Q, K, V = rope._torch_reference(qkv, nh, nkv, hd, hd, base, 1.0, cu)
        assert Q.dtype == BF
        out[f"rope{i}_qkv"], out[f"rope{i}_cfg"], out[f"rope{i}_lens"] = bits(qkv), np.array([nh, nkv, hd, base]), np.array(lens)
        out[f"rope{i}_q"], out[f"rope{i}_k"], out[f"rope{i}_v"] = bits(Q), bits(K), bits(V)
    out["rope_count"] = np.array(2)
    for i, (nh, nkv, hd, block, ctx) in enumerate([(28, 4, 128, 16, [1, 17, 70]), (8, 1, 128, 16, [5, 100])]):
        B, x = len(ctx), 8
        mb = (max(ctx) + block - 1) // block
        nblk = B * mb
        bt = torch.randperm(nblk, generator=g).reshape(B, mb).to(torch.int32)
        q = rn(B, nh, hd).to(BF)
        Kn, Vn = rn(nblk * block, nkv, hd).to(BF), rn(nblk * block, nkv, hd).to(BF)
        k_cache = Kn.view(nblk, block, nkv, hd // x, x).permute(0, 2, 3, 1, 4).contiguous()
        v_cache = Vn.view(nblk, block, nkv, hd).permute(0, 2, 3, 1).contiguous()
        one = torch.tensor([1.0])
        o = fmha.run_native(q, k_cache, v_cache, bt, torch.tensor(ctx), max(ctx), "auto", nkv, 1.0 / hd ** 0.5, None, one, one,
                            nh // nkv, BF)
        assert o.dtype == BF
        out[f"attn{i}_cfg"] = np.array([nh, nkv, hd, block])
        out[f"attn{i}_ctx"], out[f"attn{i}_bt"], out[f"attn{i}_q"] = np.array(ctx), bt.numpy(), bits(q)
        out[f"attn{i}_k"], out[f"attn{i}_v"], out[f"attn{i}_out"] = bits(Kn), bits(Vn), bits(o)
    out["attn_count"] = np.array(2)
    H, I, T = 256, 640, 9
    x, gate, up, down = rn(T, H, s=0.5).to(BF), rn(H, I, s=0.06).to(BF), rn(H, I, s=0.06).to(BF), rn(I, H, s=0.04).to(BF)
    y = mlp.DenseMLP(gate, up, down, sys.modules["rtp_llm.ops"].ActivationType.Swiglu)(x)
    out.update(mlp_x=bits(x), mlp_gate=bits(gate), mlp_up=bits(up), mlp_down=bits(down), mlp_y=bits(y))
    np.savez_compressed(os.path.join(OUT, "ref_layers_bf16.npz"), **out)
    print("wrote ref_layers_bf16.npz")


def main():
    os.makedirs(OUT, exist_ok=True)
    mod = _import_reference_device_impl()
    impl = _make_rocm_impl(mod)

    K, N, g = 256, 64, 128
    # ---------------- GPTQ
    qw, qz, sc = _rand_gptq(K, N, g, 1)
    w, zs, s = impl.preprocess_groupwise_weight_params(qw.clone(), qz.clone(), sc.clone(), "cpu", True, False, 4)
    # canonical intermediates straight from the reference helpers (device_impl.py:148-171)
    q_codes = impl.unpack_int32_into_int16(qw.T, False).T.contiguous()        # [K, N] 0..15
    z_codes = impl.unpack_int32_into_int16(qz, False)                         # [K/g, N] 0..15
    np.savez_compressed(os.path.join(OUT, "quant_gptq.npz"), qweight=qw.numpy(), qzeros=qz.numpy(), scales=sc.numpy(),
                        ref_kernel=w.contiguous().view(torch.uint8).numpy(), ref_kernel_stride=np.array(w.stride()),
                        ref_kernel_shape=np.array(w.shape), ref_zeros_x_scales=zs.contiguous().numpy(),
                        ref_scales=s.contiguous().numpy(), ref_q_codes=q_codes.numpy().astype(np.uint8),
                        ref_z_codes=z_codes.numpy().astype(np.uint8), group_size=np.array(g))
    # ---------------- AWQ
    qw, qz, sc = _rand_awq(K, N, g, 2)
    w, zs, s = impl.preprocess_groupwise_weight_params(qw.clone(), qz.clone(), sc.clone(), "cpu", False, True, 4)
    q_codes = impl.reverse_awq_order(impl.unpack_int32_into_int16(qw, False))
    z_codes = impl.reverse_awq_order(impl.unpack_int32_into_int16(qz, False))
    np.savez_compressed(os.path.join(OUT, "quant_awq.npz"), qweight=qw.numpy(), qzeros=qz.numpy(), scales=sc.numpy(),
                        ref_kernel=w.contiguous().view(torch.uint8).numpy(), ref_kernel_stride=np.array(w.stride()),
                        ref_kernel_shape=np.array(w.shape), ref_zeros_x_scales=zs.contiguous().numpy(),
                        ref_scales=s.contiguous().numpy(), ref_q_codes=q_codes.numpy().astype(np.uint8),
                        ref_z_codes=z_codes.numpy().astype(np.uint8), group_size=np.array(g))
    # ---------------- INT8 autoquant (per-channel); the ROCm preprocessor is the identity-free
    # 4-bit permutation, so quantise through the method itself with packing bypassed
    gen = torch.Generator().manual_seed(3)
    W = (torch.randn(192, 48, generator=gen) * 0.05).half()
    impl.preprocess_weights_for_mixed_gemm = lambda t, mode, arch="": t  # keep canonical int8 [K,N]
    q8, s8 = impl.apply_int8(W.clone(), "cpu")
    np.savez_compressed(os.path.join(OUT, "quant_int8.npz"), weight=W.numpy(), ref_q=q8.numpy(), ref_scale=s8.numpy())
    gen_ref_layers()
    gen_ref_layers_bf16()
    print("golden vectors written to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
