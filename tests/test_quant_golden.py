"""Pin the load-time quantisation math against golden vectors produced by the reference's own
rtp_llm/device/device_impl.py (generator: oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from rtp_llm_amd import quant

CK_PERM = [2, 0, 6, 4, 3, 1, 7, 5]  # device_impl.py:751


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _decode_reference_rocm_kernel(ref_kernel_u8: np.ndarray, K: int, N: int) -> np.ndarray:
    """Invert RocmImpl.pack_int8_tensor_to_packed_int4 + preprocess_weights_for_mixed_gemm
    (device_impl.py:729-771): logical [K/2, N] with column-major storage, per-8-nibble
    permutation CK_PERM, XOR 0x88 (offset binary = the original unsigned code)."""
    storage = np.ascontiguousarray(ref_kernel_u8.reshape(K // 2, N).T).reshape(-1)   # [N][K/2] bytes
    nib = np.stack([storage >> 4, storage & 0xF], axis=1).reshape(-1, 8)            # reordered nibbles
    orig = np.empty_like(nib)
    orig[:, CK_PERM] = nib                                                           # reordered[j] = orig[perm[j]]
    return orig.reshape(N, K).T                                                      # q[k][n]


@pytest.mark.parametrize("kind", ["gptq", "awq"])
def test_unpack_matches_reference(golden_dir, kind):
    g = _load(golden_dir, f"quant_{kind}.npz")
    qw, qz = torch.from_numpy(g["qweight"]), torch.from_numpy(g["qzeros"])
    q, z_eff = (quant.unpack_gptq if kind == "gptq" else quant.unpack_awq)(qw, qz)
    assert np.array_equal(q.numpy(), g["ref_q_codes"])
    flag = 1 if kind == "gptq" else 0
    assert np.array_equal(z_eff.numpy().astype(np.int16), g["ref_z_codes"].astype(np.int16) + flag)
    # the reference's device-packed kernel bytes decode to the same codes
    K, N = q.shape
    assert np.array_equal(_decode_reference_rocm_kernel(g["ref_kernel"], K, N), q.numpy())


@pytest.mark.parametrize("kind", ["gptq", "awq"])
def test_folded_zeros_and_dequant_match_reference(golden_dir, kind):
    g = _load(golden_dir, f"quant_{kind}.npz")
    gs = int(g["group_size"])
    qw, qz, sc = torch.from_numpy(g["qweight"]), torch.from_numpy(g["qzeros"]), torch.from_numpy(g["scales"])
    q, z_eff = (quant.unpack_gptq if kind == "gptq" else quant.unpack_awq)(qw, qz)
    zs = quant.reference_folded_zeros(z_eff, sc)
    assert np.array_equal(zs.numpy().view(np.uint16), g["ref_zeros_x_scales"].view(np.uint16))      # bit exact
    assert np.array_equal(sc.numpy().view(np.uint16), g["ref_scales"].view(np.uint16))
    # oracle dequant (exact) vs the reference's folded kernel-side representation: equal up to the
    # fp16 rounding of zeros_x_scales
    w_exact = oracle.dequant_groupwise(q, z_eff, sc, gs)
    w_ref = (torch.from_numpy(g["ref_q_codes"]).float() - 8.0) * sc.float().repeat_interleave(gs, 0) \
        + torch.from_numpy(g["ref_zeros_x_scales"]).float().repeat_interleave(gs, 0)
    assert torch.allclose(w_exact, w_ref, atol=2e-4, rtol=1e-3)
    assert torch.equal(oracle.dequant_reference_folded(q, z_eff, sc, gs), w_ref)


def test_int8_autoquant_matches_reference(golden_dir):
    g = _load(golden_dir, "quant_int8.npz")
    W = torch.from_numpy(g["weight"])
    q, s = quant.symmetric_quantize_int8(W)
    assert np.array_equal(q.numpy(), g["ref_q"])
    assert np.array_equal(s.numpy().view(np.uint16), g["ref_scale"].view(np.uint16))


def _unpack_native_w4(img: torch.Tensor, K_pad: int, N_pad: int) -> torch.Tensor:
    """Independent (loop) decoder of the native W4 tile image (include/mi355_decode.h)."""
    KC, NT = K_pad // 128, N_pad // 16
    w = (img.reshape(NT, KC, 64, 4).to(torch.int64) & 0xFFFFFFFF).numpy()
    out = np.zeros((K_pad, N_pad), dtype=np.uint8)
    for nt in range(NT):
        for c in range(KC):
            for lane in range(64):
                i, qq = lane & 15, lane >> 4
                for s in range(4):
                    for e in range(8):
                        shift = 4 * (e // 2) + 16 * (e & 1)
                        out[128 * c + 32 * s + 8 * qq + e, 16 * nt + i] = (int(w[nt, c, lane, s]) >> shift) & 0xF
    return torch.from_numpy(out)


def test_native_w4_image_roundtrip():
    torch.manual_seed(0)
    K, N = 384, 80
    q = torch.randint(0, 16, (K, N), dtype=torch.uint8)
    img = quant.pack_w4(q)
    assert img.numel() == (K // 128) * (N // 16) * 64 * 4
    assert torch.equal(_unpack_native_w4(img, K, N)[:K, :N], q)


def test_native_w8_w16_images():
    torch.manual_seed(1)
    K, N = 256, 32
    q = torch.randint(-128, 128, (K, N), dtype=torch.int16).to(torch.int8)
    img = quant.pack_w8(q).reshape(N // 16, K // 128, 2, 64, 16)
    # lane (i, qq), wave-load p, byte h*8+e  <->  k = 128c + 32(2p+h) + 8qq + e, n = 16nt + i
    for (nt, c, p, qq, i, h, e) in [(0, 0, 0, 0, 0, 0, 0), (1, 1, 1, 3, 15, 1, 7), (0, 1, 0, 2, 5, 1, 3)]:
        k, n = 128 * c + 32 * (2 * p + h) + 8 * qq + e, 16 * nt + i
        assert int(img[nt, c, p, qq * 16 + i, h * 8 + e]) == int(q[k, n]) + 128
    w = torch.randn(K, N).half()
    img16 = quant.pack_w16(w).reshape(N // 16, K // 128, 4, 64, 8)
    for (nt, c, s, qq, i, e) in [(0, 0, 0, 0, 0, 0), (1, 1, 3, 3, 15, 7), (1, 0, 2, 1, 9, 4)]:
        assert img16[nt, c, s, qq * 16 + i, e] == w[128 * c + 32 * s + 8 * qq + e, 16 * nt + i]


def test_interleave_gate_up():
    t = torch.arange(12).reshape(1, 12)
    out = quant.interleave_gate_up(t)
    assert out.tolist() == [[0, 6, 1, 7, 2, 8, 3, 9, 4, 10, 5, 11]]
    t0 = torch.arange(8).reshape(8, 1)
    assert quant.interleave_gate_up(t0, dim=0).reshape(-1).tolist() == [0, 4, 1, 5, 2, 6, 3, 7]
