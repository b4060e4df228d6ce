"""Load-time weight quantisation, canonicalisation and MI355-native repacking.

Host-side mirror of the reference's load-time half of the weight-only path
(the only part of W4A16/W8A16 that survives in the reference):

  * ``symmetric_quantize_int8``  <- GpuImpl.symmetric_quantize_last_axis_of_batched_matrix
                                    + apply_int8 (rtp_llm/device/device_impl.py:183-222)
  * ``unpack_gptq`` / ``unpack_awq`` <- unpack_int32_into_int16 / reverse_awq_order
                                    (device_impl.py:148-171) as used by
                                    preprocess_groupwise_weight_params (:242-300, ROCm :797-868)
  * ``pack_*`` / ``make_meta``   <- the per-device packing step
                                    (RocmImpl.pack_int8_tensor_to_packed_int4 +
                                    preprocess_weights_for_mixed_gemm, :729-771); the image
                                    produced here is the MI355-native tile layout described
                                    in include/mi355_decode.h, not CK's nibble permutation.

Everything is written with torch ops only, so it runs on CPU (tests) or on the
GPU (bench: 7B-sized synthetic weights are generated and packed on device).
"""
from dataclasses import dataclass
from typing import Optional

import torch

AWQ_REVERSE_ORDER = (0, 4, 1, 5, 2, 6, 3, 7)  # device_impl.py:164
# nibble e of a k-step sits at this bit of the dword (pairs (2t, 2t+1) land in one fp16x2)
_W4_SHIFTS = (0, 16, 4, 20, 8, 24, 12, 28)


def _ceil_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


# --------------------------------------------------------------------------- canonicalisation
def _unpack_nibbles_last(t_int32: torch.Tensor) -> torch.Tensor:
    """int32 [..., C] -> uint8 [..., 8C]; nibble j of word c (bits 4j..4j+3) -> column 8c + j."""
    shifts = torch.arange(0, 32, 4, device=t_int32.device, dtype=torch.int32)
    x = (t_int32.unsqueeze(-1) >> shifts) & 0xF
    return x.reshape(*t_int32.shape[:-1], t_int32.shape[-1] * 8).to(torch.uint8)


def _undo_awq_order(t: torch.Tensor) -> torch.Tensor:
    """logical column 8c + j = packed column 8c + AWQ_REVERSE_ORDER[j]."""
    idx = torch.tensor(AWQ_REVERSE_ORDER, device=t.device)
    return t.reshape(*t.shape[:-1], -1, 8)[..., idx].reshape(t.shape)


def unpack_gptq(qweight: torch.Tensor, qzeros: torch.Tensor):
    """AutoGPTQ 4-bit tensors -> (q uint8 [K,N] in 0..15, z_eff uint8 [K/g,N]).

    qweight int32 [K/8, N] packs 8 consecutive k per word (low nibble first);
    qzeros int32 [K/g, N/8] packs along N.  W = scale * (q - (z + 1)): GPTQ_FLAG = 1
    (device_impl.py:252,285-287).
    """
    assert qweight.dtype == torch.int32 and qzeros.dtype == torch.int32
    q = _unpack_nibbles_last(qweight.t().contiguous()).t().contiguous()  # [K, N]
    z = _unpack_nibbles_last(qzeros.contiguous())                        # [K/g, N]
    return q, (z.to(torch.int16) + 1).to(torch.uint8)


def unpack_awq(qweight: torch.Tensor, qzeros: torch.Tensor):
    """AutoAWQ 4-bit tensors -> (q uint8 [K,N], z_eff uint8 [K/g,N]).

    qweight int32 [K, N/8] and qzeros int32 [K/g, N/8] pack along N in the order
    [0,2,4,6,1,3,5,7] (undone by reverse_awq_order, device_impl.py:163-171).  W = scale * (q - z).
    """
    assert qweight.dtype == torch.int32 and qzeros.dtype == torch.int32
    q = _undo_awq_order(_unpack_nibbles_last(qweight.contiguous()))
    z = _undo_awq_order(_unpack_nibbles_last(qzeros.contiguous()))
    return q, z


def symmetric_quantize_int8(weight: torch.Tensor):
    """Load-time INT8 autoquant, per output column (reference a1, device_impl.py:183-192):
    scale[n] = max(|W[:, n]|.max(), 1e-8) / 128;  q = clamp(round(W / scale), -128, 127).
    weight: [K, N] (in, out).  Returns (q int8 [K,N], scale [N] in weight.dtype)."""
    amax = weight.abs().max(dim=0)[0]
    amax = torch.clamp(amax, min=1e-8)
    scale = amax / 128.0
    q = torch.clamp((weight / scale).round(), -128, 127).to(torch.int8)
    return q, scale


# --------------------------------------------------------------------------- native images
def _pad2(t: torch.Tensor, K_pad: int, N_pad: int, value=0) -> torch.Tensor:
    K, N = t.shape
    if K == K_pad and N == N_pad:
        return t
    out = torch.full((K_pad, N_pad), value, dtype=t.dtype, device=t.device)
    out[:K, :N] = t
    return out


def pack_w4(q: torch.Tensor) -> torch.Tensor:
    """q uint8 [K,N] codes 0..15 -> int32 image [NT, KC, 64 lanes, 4 dwords] (flattened)."""
    K, N = q.shape
    K_pad, N_pad = _ceil_to(K, 128), _ceil_to(N, 16)
    q = _pad2(q, K_pad, N_pad)
    KC, NT = K_pad // 128, N_pad // 16
    x = q.reshape(KC, 4, 4, 8, NT, 16)            # [c, s, qq, e, nt, i]
    x = x.permute(4, 0, 2, 5, 1, 3).to(torch.int64)  # [nt, c, qq, i, s, e]
    shifts = torch.tensor(_W4_SHIFTS, device=q.device, dtype=torch.int64)
    w = (x << shifts).sum(dim=-1)                  # [nt, c, qq, i, s]
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    return w.reshape(-1).contiguous()


def pack_w8(q: torch.Tensor) -> torch.Tensor:
    """q int8 [K,N] -> uint8 image [NT, KC, 2, 64 lanes, 16 bytes] of offset-binary codes (q+128)."""
    K, N = q.shape
    K_pad, N_pad = _ceil_to(K, 128), _ceil_to(N, 16)
    u = (q.to(torch.int16) + 128).to(torch.uint8)
    u = _pad2(u, K_pad, N_pad, 128)
    KC, NT = K_pad // 128, N_pad // 16
    x = u.reshape(KC, 2, 2, 4, 8, NT, 16)          # [c, p, h, qq, e, nt, i]
    x = x.permute(5, 0, 1, 3, 6, 2, 4)             # [nt, c, p, qq, i, h, e]
    return x.reshape(-1).contiguous()


def pack_w16(w: torch.Tensor) -> torch.Tensor:
    """fp16 / bf16 [K,N] -> image of the same dtype [NT, KC, 4 steps, 64 lanes, 8 elements]."""
    assert w.dtype in (torch.float16, torch.bfloat16)
    K, N = w.shape
    K_pad, N_pad = _ceil_to(K, 128), _ceil_to(N, 16)
    w = _pad2(w, K_pad, N_pad)
    KC, NT = K_pad // 128, N_pad // 16
    x = w.reshape(KC, 4, 4, 8, NT, 16)             # [c, s, qq, e, nt, i]
    x = x.permute(4, 0, 1, 2, 5, 3)                # [nt, c, s, qq, i, e]
    return x.reshape(-1).contiguous()


def make_meta(z_eff: torch.Tensor, scales: torch.Tensor, N_pad: int) -> torch.Tensor:
    """z_eff [G,N] integer codes, scales [G,N] -> int32 [G, N_pad] of fp16x2 {-(1024+z_eff), scale}."""
    G, N = scales.shape
    zneg = (-(1024.0 + z_eff.to(torch.float32))).to(torch.float16)
    sc = scales.to(torch.float16)
    m = torch.zeros(G, N_pad, 2, dtype=torch.float16, device=scales.device)
    m[:, :N, 0] = zneg
    m[:, :N, 1] = sc
    m[:, N:, 0] = -1024.0
    return m.view(torch.int32).reshape(G, N_pad).contiguous()


def interleave_gate_up(t: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """[gate | up] halves along `dim` -> (g0,u0,g1,u1,...): the column order the fused
    SiLU-gate GEMM epilogue expects (each lane then owns (gate, up) pairs)."""
    dim = dim % t.dim()
    n = t.shape[dim]
    g, u = t.narrow(dim, 0, n // 2), t.narrow(dim, n // 2, n // 2)
    return torch.stack((g, u), dim=dim + 1).reshape(*t.shape[:dim], n, *t.shape[dim + 1:]).contiguous()


# --------------------------------------------------------------------------- packed weight object
@dataclass
class PackedWeight:
    """Python twin of mi355_weight_t (include/mi355_decode.h)."""
    qweight: torch.Tensor           # native image
    meta: Optional[torch.Tensor]    # int32 [G, N_pad] or None (W16)
    wbits: int
    K: int
    N: int
    K_pad: int
    N_pad: int
    group_size: int                 # 0 = per-channel

    def to(self, device):
        return PackedWeight(self.qweight.to(device), None if self.meta is None else self.meta.to(device),
                            self.wbits, self.K, self.N, self.K_pad, self.N_pad, self.group_size)

    @property
    def nbytes(self) -> int:
        return self.qweight.numel() * self.qweight.element_size() + (
            0 if self.meta is None else self.meta.numel() * 4)


def pack_groupwise_w4(q: torch.Tensor, z_eff: torch.Tensor, scales: torch.Tensor, group_size: int) -> PackedWeight:
    K, N = q.shape
    assert group_size in (32, 64, 128), "group_size must be 32, 64 or 128"
    assert K % group_size == 0 and scales.shape == (K // group_size, N) and z_eff.shape == scales.shape
    K_pad, N_pad = _ceil_to(K, 128), _ceil_to(N, 16)
    G_pad = K_pad // group_size
    if G_pad != scales.shape[0]:  # zero-scale groups for the K padding
        pad = G_pad - scales.shape[0]
        scales = torch.cat([scales, torch.zeros(pad, N, dtype=scales.dtype, device=scales.device)])
        z_eff = torch.cat([z_eff, torch.zeros(pad, N, dtype=z_eff.dtype, device=z_eff.device)])
    return PackedWeight(pack_w4(q), make_meta(z_eff, scales, N_pad), 4, K, N, K_pad, N_pad, group_size)


def pack_gptq(qweight, qzeros, scales, group_size=128) -> PackedWeight:
    q, z = unpack_gptq(qweight, qzeros)
    return pack_groupwise_w4(q, z, scales, group_size)


def pack_awq(qweight, qzeros, scales, group_size=128) -> PackedWeight:
    q, z = unpack_awq(qweight, qzeros)
    return pack_groupwise_w4(q, z, scales, group_size)


def pack_int8_per_channel(q: torch.Tensor, scale: torch.Tensor) -> PackedWeight:
    K, N = q.shape
    K_pad, N_pad = _ceil_to(K, 128), _ceil_to(N, 16)
    z = torch.full((1, N), 128, dtype=torch.int16, device=q.device)
    return PackedWeight(pack_w8(q), make_meta(z, scale.reshape(1, N), N_pad), 8, K, N, K_pad, N_pad, 0)


def pack_fp16(w: torch.Tensor) -> PackedWeight:
    K, N = w.shape
    return PackedWeight(pack_w16(w), None, 16, K, N, _ceil_to(K, 128), _ceil_to(N, 16), 0)


def autoquant_int8(weight_kn: torch.Tensor) -> PackedWeight:
    """fp16 [K,N] -> W8A16 packed weight (``--quantization int8`` load-time autoquant)."""
    q, s = symmetric_quantize_int8(weight_kn)
    return pack_int8_per_channel(q, s)


# --------------------------------------------------------------------------- reference-format import
CK_NIBBLE_PERM = (2, 0, 6, 4, 3, 1, 7, 5)   # device_impl.py:751


def unpack_reference_rocm_w4(kernel: torch.Tensor) -> torch.Tensor:
    """The tensor the reference's ROCm loader hands the linear factory as ``W.*_w`` for a 4-bit layer -> canonical codes.

    RocmImpl.preprocess_groupwise_weight_params (device_impl.py:797-868) emits int8 [K/2, N] (column-major strides): two
    codes per byte along K (even k in the high nibble), XOR 0x88 (so the byte holds the ORIGINAL unsigned codes),
    then the CK nibble permutation [2,0,6,4,3,1,7,5] per 8 nibbles of the column-major byte stream (device_impl.py:729-771).
    Returns q uint8 [K, N] in 0..15."""
    assert kernel.dtype == torch.int8 and kernel.dim() == 2
    K2, N = kernel.shape
    storage = kernel.t().contiguous().view(torch.uint8).reshape(-1)           # bytes in column-major order: [N][K/2]
    nib = torch.stack([storage >> 4, storage & 0xF], dim=1).reshape(-1, 8)    # permuted nibbles
    orig = torch.empty_like(nib)
    orig[:, list(CK_NIBBLE_PERM)] = nib                                       # permuted[j] = orig[perm[j]]
    return orig.reshape(N, 2 * K2).t().contiguous()


def zeros_from_reference_folded(zeros_x_scales: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """Invert zeros_x_scales = fp16((8 - z_eff) * scale) (device_impl.py:836-840): the quotient is within 2^-11 of an
    integer, so rounding recovers z_eff exactly; zero-scale groups carry no information (weights are 0): z_eff = 8."""
    s = scales.float()
    ratio = torch.where(s != 0, zeros_x_scales.float() / torch.where(s != 0, s, torch.ones_like(s)), torch.zeros_like(s))
    z = 8 - torch.round(ratio)
    assert bool(((z >= 0) & (z <= 16)).all()), "zeros_x_scales is not (8 - z) * scale for integer z in 0..16"
    return z.to(torch.uint8)


def pack_reference_rocm_w4(kernel: torch.Tensor, scales: torch.Tensor, zeros_x_scales: torch.Tensor) -> "PackedWeight":
    """(W.*_w, W.*_s, W.*_z) exactly as the reference's ROCm loader produces them -> the MI355-native image; bit-equal to
    pack_gptq / pack_awq on the checkpoint tensors they came from."""
    q = unpack_reference_rocm_w4(kernel)
    K = q.shape[0]
    G = scales.shape[0]
    assert K % G == 0 and K // G in (32, 64, 128), f"group size {K}/{G}"
    return pack_groupwise_w4(q, zeros_from_reference_folded(zeros_x_scales, scales), scales.contiguous().to(torch.float16), K // G)


def cat_groupwise_cols(parts):
    """Column-concatenate (weight-codes, scales, zeros) triples -- the scale AND zero concat the reference's
    create_merged_linear lacks for GPTQ/AWQ (it concatenates scales for FP8/FP4 only, factory.py:172-176)."""
    return tuple(torch.cat([p[i] for p in parts], dim=-1) for i in range(3))


# --------------------------------------------------------------------------- reference-format export
def reference_folded_zeros(z_eff: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """The reference's kernel-side representation (device_impl.py:283-289):
    zeros_x_scales = (8 - z - GPTQ_FLAG) * scale = (8 - z_eff) * scale, fp16,
    used with signed nibbles q_s = q - 8:  W = q_s * scale + zeros_x_scales."""
    return ((8 - z_eff.to(torch.int16)).to(scales.dtype) * scales).half()
