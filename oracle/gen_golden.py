"""Generate golden vectors by running the REFERENCE's own load-time quantisation code.

TEST INFRASTRUCTURE ONLY.  Runs in the authoring container only (needs /root/reference);
the committed outputs (tests/golden/quant_*.npz) are what travels.

The reference's ``rtp_llm/device/device_impl.py`` is pure Python over torch once its
imports of the compiled op libraries are stubbed (verified importable with
``sys.modules`` stubs).  We execute, unmodified:
  * RocmImpl.preprocess_groupwise_weight_params (GPTQ and AWQ, 4-bit)  device_impl.py:797-868
  * GpuImpl.apply_int8 / symmetric_quantize_last_axis_of_batched_matrix  device_impl.py:183-222
and record inputs + outputs.  From the ROCm-packed int4 output we also recover the canonical
codes (undoing the CK nibble permutation and column-major packing, device_impl.py:729-771)
so tests can pin rtp_llm_amd.quant.unpack_gptq/unpack_awq bit-exactly.

    python oracle/gen_golden.py            # writes tests/golden/quant_{gptq,awq,int8}.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _import_reference_device_impl():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; golden vectors are committed under tests/golden/")

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

    for name in ["rtp_llm", "rtp_llm.ops", "rtp_llm.ops.compute_ops", "rtp_llm.config", "rtp_llm.config.py_config_modules",
                 "rtp_llm.utils", "rtp_llm.utils.model_weight", "rtp_llm.utils.swizzle_utils", "rtp_llm.device",
                 "rtp_llm.device.device_type"]:
        sys.modules[name] = _Stub(name)
    if "psutil" not in sys.modules:
        try:
            import psutil  # noqa: F401
        except ImportError:
            sys.modules["psutil"] = _Stub("psutil")

    def load(modname, rel):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    load("rtp_llm.device.device_base", "rtp_llm/device/device_base.py")
    return load("rtp_llm.device.device_impl", "rtp_llm/device/device_impl.py")


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
    print("golden vectors written to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
