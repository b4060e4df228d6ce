"""Tensor-level wrappers over the C-ABI: the counterpart of the reference's pybind op
module ``rtp_llm.ops.compute_ops.rtp_llm_ops`` (bindings/rocm/RegisterBaseBindings.hpp:14-129)
for the decode hot path.  PyTorch is plumbing here (device memory + current stream);
every function enqueues hand-written HIP kernels through ctypes and never falls back.
"""
import ctypes as C
import math
from typing import Optional

import torch

from . import _C
from .quant import PackedWeight


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _C.Mi355Error(f"{name}: tensor must live on the GPU (no CPU path exists)")
    if t.dtype != dtype:
        raise _C.Mi355Error(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _C.Mi355Error(f"{name}: tensor must be contiguous")


ACT_DTYPES = (torch.float16, torch.bfloat16)


def _chk_act(t: torch.Tensor, name: str, like: Optional[torch.Tensor] = None):
    """An activation tensor: fp16 or bf16 (the reference's dtype grid), on the GPU, contiguous; `like` fixes the dtype of a call."""
    if t.dtype not in ACT_DTYPES:
        raise _C.Mi355Error(f"{name}: expected float16 or bfloat16, got {t.dtype}")
    _chk(t, t.dtype if like is None else like.dtype, name)


def weight_struct(w: PackedWeight, act: torch.dtype = torch.float16) -> _C.Weight:
    if w.wbits == 16 and w.qweight.dtype != act:
        raise _C.Mi355Error(f"linear: 16-bit weight image is {w.qweight.dtype}, activations are {act}")
    return _C.Weight(w.qweight.data_ptr(), 0 if w.meta is None else w.meta.data_ptr(), w.wbits, w.K, w.N,
                     w.K_pad, w.N_pad, w.group_size, _C.ACT_BF16 if act == torch.bfloat16 else _C.ACT_F16)


def kv_struct(kv_base: torch.Tensor, scale_base: Optional[torch.Tensor], page: int, nkv: int, hd: int,
              act: Optional[torch.dtype] = None) -> _C.KVLayer:
    """act: dtype of the Q / K / V rows that go with this cache (default: the cache's own 16-bit dtype, fp16 for an INT8 cache)."""
    int8 = kv_base.dtype == torch.int8
    if int8 and scale_base is None:
        raise _C.Mi355Error("int8 KV cache requires the fp32 scale plane")
    nblk = kv_base.numel() // (2 * nkv * page * hd)
    kvd = _C.KV_INT8 if int8 else (_C.KV_BF16 if kv_base.dtype == torch.bfloat16 else _C.KV_FP16)
    act = _act_of_cache(kv_base) if act is None else act
    if not int8 and act != kv_base.dtype:
        raise _C.Mi355Error(f"a {kv_base.dtype} cache takes {kv_base.dtype} rows, got {act}")
    return _C.KVLayer(kv_base.data_ptr(), 0 if scale_base is None else scale_base.data_ptr(), kvd, page, nkv, hd, nblk,
                      _C.ACT_BF16 if act == torch.bfloat16 else _C.ACT_F16)


def _act_of_cache(kv_base: torch.Tensor) -> torch.dtype:
    """Q / K / V / attention-output dtype that goes with a cache: bf16 cache <-> bf16 activations; fp16 and INT8 caches <-> fp16."""
    return torch.bfloat16 if kv_base.dtype == torch.bfloat16 else torch.float16


def _dt(t: torch.Tensor) -> int:
    return _C.ACT_BF16 if t.dtype == torch.bfloat16 else _C.ACT_F16


# ------------------------------------------------------------------ linear
_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def linear(x: torch.Tensor, w: PackedWeight, bias: Optional[torch.Tensor] = None, epilogue: int = _C.EPI_NONE,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ W (+bias); epilogue EPI_SILU_MUL -> [M, N/2], EPI_OUT_F32 -> fp32.  x fp16 or bf16 (bias and y follow)."""
    _chk_act(x, "linear.x")
    if bias is not None:
        _chk_act(bias, "linear.bias", x)
    M = x.numel() // w.K
    if x.shape[-1] != w.K:
        raise _C.Mi355Error(f"linear: x last dim {x.shape[-1]} != K {w.K}")
    n_out = w.N // 2 if epilogue & _C.EPI_SILU_MUL else w.N   # (HINT_* bits of `epilogue` only select the kernel family)
    dt = torch.float32 if epilogue & _C.EPI_OUT_F32 else x.dtype
    if out is None:
        out = torch.empty(*x.shape[:-1], n_out, dtype=dt, device=x.device)
    ws_struct = weight_struct(w, x.dtype)
    need = _C.lib().mi355_linear_workspace_bytes(M, C.byref(ws_struct))
    ws = _workspace(need, x.device)
    _C.check(_C.lib().mi355_linear_forward(x.data_ptr(), M, C.byref(ws_struct), _p(bias), out.data_ptr(), epilogue,
                                           ws.data_ptr(), ws.numel(), _stream()), "linear_forward")
    return out


ERR_UNSUPPORTED = -3


def _fused_norm(norm, M: int, K: int):
    """(tile_sumsq [rows >= M, ld >= K/16] fp32, weight [K] fp16, eps) -> mi355_fused_norm_t (kept alive by the caller's tuple)."""
    if norm is None:
        return None
    tile_sumsq, weight, eps = norm
    _chk(tile_sumsq, torch.float32, "fused norm tile_sumsq"); _chk(weight, torch.float16, "fused norm weight")
    if tile_sumsq.dim() != 2 or tile_sumsq.shape[0] < M or tile_sumsq.shape[1] < K // 16 or weight.numel() != K:
        raise _C.Mi355Error(f"fused norm: tile_sumsq {tuple(tile_sumsq.shape)} / weight {weight.numel()} against K={K} M={M}")
    return _C.FusedNorm(tile_sumsq.data_ptr(), K // 16, tile_sumsq.shape[1], weight.data_ptr(), float(eps))


def linear_residual(x: torch.Tensor, w: PackedWeight, residual: torch.Tensor, bias: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None, tile_sumsq: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """residual + fp16(x @ W + bias) in one launch (gemm_fullk.hip); None when the shape is not taken by the fused kernel
    (the caller composes linear + add).  `out` may alias `residual`.  tile_sumsq ([rows >= M, ld >= N/16] fp32) receives the per-tile
    sums of squares of the produced rows -- the input of a consumer's fused RMSNorm."""
    _chk(x, torch.float16, "linear_residual.x"); _chk(residual, torch.float16, "linear_residual.residual")
    M = x.numel() // w.K
    if x.shape[-1] != w.K or residual.shape[-1] != w.N or residual.numel() != M * w.N:
        raise _C.Mi355Error(f"linear_residual: x {tuple(x.shape)} / residual {tuple(residual.shape)} against K={w.K} N={w.N}")
    if tile_sumsq is not None:
        _chk(tile_sumsq, torch.float32, "linear_residual.tile_sumsq")
        if tile_sumsq.dim() != 2 or tile_sumsq.shape[0] < M or tile_sumsq.shape[1] < w.N // 16:
            raise _C.Mi355Error(f"linear_residual: tile_sumsq {tuple(tile_sumsq.shape)} must be [>= {M}, >= N/16 = {w.N // 16}]")
    if out is None:
        out = torch.empty_like(residual)
    ws_struct = weight_struct(w)
    rc = _C.lib().mi355_linear_residual(x.data_ptr(), M, C.byref(ws_struct), _p(bias), residual.data_ptr(), out.data_ptr(),
                                        _p(tile_sumsq), 0 if tile_sumsq is None else tile_sumsq.shape[1], _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "linear_residual")
    return out


def norm_linear(h: torch.Tensor, norm, w: PackedWeight, bias: Optional[torch.Tensor] = None, epilogue: int = _C.EPI_NONE):
    """epilogue(RMSNorm(h) @ W + bias) in one launch; norm = (tile_sumsq, weight, eps) as left by linear_residual.  None when
    not taken (more than 16 rows, non-W4 weights)."""
    _chk(h, torch.float16, "norm_linear.h")
    if norm is None:
        raise _C.Mi355Error("norm_linear needs norm = (tile_sumsq, weight, eps) as left by linear_residual")
    M = h.numel() // w.K
    fn = _fused_norm(norm, M, w.K)
    n_out = w.N // 2 if epilogue & _C.EPI_SILU_MUL else w.N
    out = torch.empty(*h.shape[:-1], n_out, dtype=torch.float32 if epilogue & _C.EPI_OUT_F32 else torch.float16, device=h.device)
    ws_struct = weight_struct(w)
    rc = _C.lib().mi355_norm_linear(h.data_ptr(), M, C.byref(fn), C.byref(ws_struct), _p(bias), out.data_ptr(), epilogue, _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "norm_linear")
    return out


def qkv_rope_kv_write(x: torch.Tensor, wqkv: PackedWeight, qkv_bias, cos_sin, positions, block_table, kv_base, scale_base,
                      nh: int, nkv: int, hd: int, page: int, q_len: int = 1, oob_count: Optional[torch.Tensor] = None, norm=None):
    """[RMSNorm +] QKV projection + bias + RoPE + Q extract + fp16 paged KV write in one launch; None when not taken (INT8
    cache, non-W4 weights): compose linear + rope_kv_write_rows then.  norm = (tile_sumsq, weight, eps): x is un-normed."""
    _chk(x, torch.float16, "qkv_rope_kv_write.x"); _chk(positions, torch.int32, "positions"); _chk(block_table, torch.int32, "block_table")
    T = x.numel() // wqkv.K
    q_out = torch.empty(T, nh, hd, dtype=torch.float16, device=x.device)
    kv = kv_struct(kv_base, scale_base, page, nkv, hd)
    ws_struct = weight_struct(wqkv)
    fn = _fused_norm(norm, T, wqkv.K)
    rc = _C.lib().mi355_qkv_rope_kv_write(x.data_ptr(), T, C.byref(ws_struct), _p(qkv_bias), None if fn is None else C.byref(fn),
                                          cos_sin.data_ptr(), hd, cos_sin.shape[0], positions.data_ptr(), block_table.data_ptr(),
                                          block_table.shape[1], q_len, nh, C.byref(kv), q_out.data_ptr(), _p(oob_count), _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "qkv_rope_kv_write")
    return q_out


# ------------------------------------------------------------------ norms / elementwise
# ---- activation images (5-64-row steps): see include/mi355_decode.h, mi355_act_image_*
class ActImage:
    """The [M][K] activations of a 5-64-row step in the order the full-K launches read them (one dense 1 KB run per MFMA fragment)."""
    def __init__(self, data: torch.Tensor, M: int, K: int, src: torch.dtype = torch.float16):
        """src: dtype of the tensor the image stands for.  The data is always fp16; the image of a bf16 tensor holds x 2^-8, saturated
        (csrc/common.h img_val: bf16 activations may exceed the fp16 range), and the GEMMs reading it scale their accumulators back."""
        self.data, self.M, self.K, self.src = data, M, K, src

    def unpack(self, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """The row-major [M, K] tensor in the image's source dtype (or converted to `dtype`)."""
        out = torch.empty(self.M, self.K, dtype=self.src, device=self.data.device)
        _C.check(_C.lib().mi355_act_image_pack(self.data.data_ptr(), self.M, self.K, out.data_ptr(), 1, _dt(out), _stream()), "act_image_pack")
        return out if dtype is None or dtype == self.src else out.to(dtype)


def _new_image(M: int, K: int, dtype, device, src: torch.dtype = torch.float16) -> ActImage:
    n = _C.lib().mi355_act_image_bytes(M, K) // 2
    return ActImage(torch.zeros(n, dtype=dtype, device=device), M, K, src)


def act_image_pack(x: torch.Tensor) -> ActImage:
    """Row-major [M, K] -> image.  Images hold fp16 (the GEMMs that read them run fp16 MFMAs): bf16 rows are stored as x 2^-8
    (ActImage.src)."""
    _chk_act(x, "act_image_pack.x")
    M, K = x.shape
    img = _new_image(M, K, torch.float16, x.device, x.dtype)
    _C.check(_C.lib().mi355_act_image_pack(x.data_ptr(), M, K, img.data.data_ptr(), 0, _dt(x), _stream()), "act_image_pack")
    return img


def add_rmsnorm_img(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, eps: float, bias: Optional[torch.Tensor] = None):
    """(y_img, residual_out): add_rmsnorm (rmsnorm when residual is None) with y written as an activation image."""
    _chk_act(x, "add_rmsnorm_img.x"); _chk_act(weight, "add_rmsnorm_img.weight", x)
    M, H = x.shape
    img = _new_image(M, H, torch.float16, x.device, x.dtype)  # fp16 data whatever the dtype of x (a bf16 result is converted on the way in)
    res_out = torch.empty_like(x) if residual is not None else None
    _C.check(_C.lib().mi355_add_rmsnorm_img(x.data_ptr(), None, 0, 0, _p(bias), _p(residual), _p(res_out), weight.data_ptr(), eps, M, H,
                                            img.data.data_ptr(), _dt(x), _stream()), "add_rmsnorm_img")
    return img, res_out


def paged_attention_rows_img(q: torch.Tensor, kv_base, scale_base, block_table: torch.Tensor, positions: torch.Tensor, nkv: int,
                             page: int, q_len: int, max_seq_len: int, scale: Optional[float] = None) -> ActImage:
    """paged_attention_rows with the output written as an activation image (<= 64 rows)."""
    _chk_act(q, "paged_attention_rows_img.q"); _chk(block_table, torch.int32, "block_table"); _chk(positions, torch.int32, "positions")
    T, nh, hd = q.shape
    kv = kv_struct(kv_base, scale_base, page, nkv, hd, q.dtype)
    img = _new_image(T, nh * hd, torch.float16, q.device, q.dtype)
    need = _C.lib().mi355_paged_attn_workspace_bytes(T, nh, hd, max_seq_len)
    ws = _workspace(need, q.device)
    _C.check(_C.lib().mi355_paged_attn_rows_img(q.data_ptr(), C.byref(kv), block_table.data_ptr(), block_table.shape[1], positions.data_ptr(),
                                                T // q_len, q_len, nh, scale if scale is not None else 1.0 / math.sqrt(hd), max_seq_len,
                                                img.data.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "paged_attn_rows_img")
    return img


def linear_residual_img(x: ActImage, w: PackedWeight, residual: torch.Tensor, bias: Optional[torch.Tensor] = None,
                        out: Optional[torch.Tensor] = None, tile_sumsq: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """linear_residual for 1-64 rows (the step driver: from 5) with the activations as an image (gemm_fullk64.hip); None when the shape is not taken."""
    _chk(x.data, torch.float16, "linear_residual_img.x"); _chk_act(residual, "linear_residual_img.residual")      # the image is fp16; residual / bias: fp16 or bf16
    M = x.M
    if x.K != w.K or residual.shape[-1] != w.N or residual.numel() != M * w.N:
        raise _C.Mi355Error(f"linear_residual_img: image {M} x {x.K} / residual {tuple(residual.shape)} against K={w.K} N={w.N}")
    if tile_sumsq is not None:
        _chk(tile_sumsq, torch.float32, "linear_residual_img.tile_sumsq")
        if tile_sumsq.dim() != 2 or tile_sumsq.shape[0] < M or tile_sumsq.shape[1] < w.N // 16:
            raise _C.Mi355Error(f"linear_residual_img: tile_sumsq {tuple(tile_sumsq.shape)} must be [>= {M}, >= N/16 = {w.N // 16}]")
    if out is None:
        out = torch.empty_like(residual)
    ws_struct = weight_struct(w, residual.dtype)
    rc = _C.lib().mi355_linear_residual_img(x.data.data_ptr(), M, C.byref(ws_struct), _p(bias), residual.data_ptr(), out.data_ptr(),
                                            _p(tile_sumsq), 0 if tile_sumsq is None else tile_sumsq.shape[1], _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "linear_residual_img")
    return out


def linear_publish_img(x: ActImage, w: PackedWeight, ar, bias: Optional[torch.Tensor] = None, dtype: torch.dtype = torch.float16) -> bool:
    """Row-parallel TP shard of a 1-64-row step: y = 16-bit(xW + bias) written by the full-K launch straight into the registered buffer of the all-reduce
    context `ar` (distributed.CustomAllReduce), for its next all_reduce_published_add_rmsnorm call.  False when the shape / format / protocol is not taken
    (mi355_linear_publish_img, include/mi355_decode.h)."""
    _chk(x.data, torch.float16, "linear_publish_img.x")
    if x.K != w.K:
        raise _C.Mi355Error(f"linear_publish_img: image {x.M} x {x.K} against K={w.K}")
    ws_struct = weight_struct(w, dtype)
    rc = _C.lib().mi355_linear_publish_img(x.data.data_ptr(), x.M, C.byref(ws_struct), _p(bias), ar.handle, _stream())
    if rc == ERR_UNSUPPORTED:
        return False
    _C.check(rc, "linear_publish_img")
    return True


def norm_exponent(weight: torch.Tensor) -> int:
    """e >= log2(max |weight|), e >= 0: the deferred RMSNorm stores weight * 2^-e * h, so that |stored| <= |h|."""
    mx = float(weight.float().abs().max())
    e = 0
    while e < 14 and 2.0 ** e < mx:
        e += 1
    return e


def linear_residual_prenorm_img(x: ActImage, w: PackedWeight, residual: torch.Tensor, norm_weight: torch.Tensor,
                                bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """(residual_out, xg_img, tile_sumsq, norm_exp): linear_residual_img that also leaves the operands of the deferred RMSNorm of the
    rows it produces (mi355_deferred_norm_t); None when the shape is not taken."""
    _chk(x.data, torch.float16, "linear_residual_prenorm_img.x"); _chk_act(residual, "residual"); _chk_act(norm_weight, "norm_weight", residual)
    M = x.M
    if x.K != w.K or residual.shape[-1] != w.N or residual.numel() != M * w.N or norm_weight.numel() != w.N:
        raise _C.Mi355Error(f"linear_residual_prenorm_img: image {M} x {x.K} / residual {tuple(residual.shape)} against K={w.K} N={w.N}")
    if out is None:
        out = torch.empty_like(residual)
    ld = (w.N // 16 + 3) & ~3
    ssq = torch.zeros(M, ld, dtype=torch.float32, device=residual.device)
    xg = _new_image(M, w.N, torch.float16, residual.device)
    e = norm_exponent(norm_weight) + (8 if residual.dtype == torch.bfloat16 else 0)     # bf16 residual stream: headroom for the fp16 image
    e = min(e, 14)
    ws_struct = weight_struct(w, residual.dtype)
    rc = _C.lib().mi355_linear_residual_prenorm_img(x.data.data_ptr(), M, C.byref(ws_struct), _p(bias), residual.data_ptr(), out.data_ptr(),
                                                    norm_weight.data_ptr(), e, xg.data.data_ptr(), ssq.data_ptr(), ld, _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "linear_residual_prenorm_img")
    return out, xg, ssq, e


def linear_deferred_norm_img(xg: ActImage, dn, w: PackedWeight, bias: Optional[torch.Tensor] = None, epilogue: int = _C.EPI_NONE,
                             act: torch.dtype = torch.float16):
    """epilogue(rs * (xg @ W) + bias) on the wide GEMM; dn = (tile_sumsq, eps, norm_exp) as left by linear_residual_prenorm_img, or None
    (plain linear on an image).  None when the shape is not taken (N too narrow to fill the chip in one launch)."""
    _chk(xg.data, torch.float16, "linear_deferred_norm_img.x")
    if xg.K != w.K:
        raise _C.Mi355Error(f"linear_deferred_norm_img: image K={xg.K} against K={w.K}")
    st = None
    if dn is not None:
        ssq, eps, e = dn
        _chk(ssq, torch.float32, "linear_deferred_norm_img.tile_sumsq")
        st = _C.DeferredNorm(ssq.data_ptr(), w.K // 16, ssq.shape[1], float(eps), float(2.0 ** e))
    N_out = w.N // 2 if (epilogue & _C.EPI_SILU_MUL) else w.N
    if epilogue & _C.EPI_OUT_IMAGE:          # the output as an activation image (the input of linear_partial_img)
        yi = _new_image(xg.M, N_out, torch.float16, xg.data.device, act)
        y = yi.data
    else:
        yi = None
        y = torch.empty(xg.M, N_out, dtype=torch.float32 if (epilogue & _C.EPI_OUT_F32) else torch.float16, device=xg.data.device)
    ws_struct = weight_struct(w, act)          # act = bfloat16: the tensors around the GEMM are bf16 (image output only, see the header)
    rc = _C.lib().mi355_linear_deferred_norm_img(xg.data.data_ptr(), xg.M, None if st is None else C.byref(st), C.byref(ws_struct), _p(bias),
                                                 y.data_ptr(), epilogue, _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "linear_deferred_norm_img")
    return yi if yi is not None else y


def linear_direct_img(x: ActImage, w: PackedWeight, bias: Optional[torch.Tensor] = None, epilogue: int = _C.EPI_NONE):
    """epilogue(x @ W + bias) in one launch from an image for a narrow N (a column-parallel TP shard; gemm_splitk64.hip, direct form); None when no plan exists.
    The tensors around the GEMM have the image's source dtype."""
    _chk(x.data, torch.float16, "linear_direct_img.x")
    if x.K != w.K:
        raise _C.Mi355Error(f"linear_direct_img: image K={x.K} against K={w.K}")
    N_out = w.N // 2 if (epilogue & _C.EPI_SILU_MUL) else w.N
    if epilogue & _C.EPI_OUT_IMAGE:
        yi = _new_image(x.M, N_out, torch.float16, x.data.device, x.src)
        y = yi.data
    else:
        yi = None
        y = torch.empty(x.M, N_out, dtype=torch.float32 if (epilogue & _C.EPI_OUT_F32) else x.src, device=x.data.device)
    ws_struct = weight_struct(w, x.src)
    rc = _C.lib().mi355_linear_direct_img(x.data.data_ptr(), x.M, C.byref(ws_struct), _p(bias), y.data_ptr(), epilogue, _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "linear_direct_img")
    return yi if yi is not None else y


def linear_partial_img(x: ActImage, w: PackedWeight, max_splits: int = 16):
    """fp32 split-K slabs [n, M, N_pad] of a deep-K linear from an activation image (gemm_splitk64.hip); None when not taken."""
    _chk(x.data, torch.float16, "linear_partial_img.x")
    if x.K != w.K:
        raise _C.Mi355Error(f"linear_partial_img: image K={x.K} against K={w.K}")
    slabs = torch.empty(max_splits, x.M, w.N_pad, dtype=torch.float32, device=x.data.device)
    ws_struct = weight_struct(w, x.src)          # the image of a bf16 tensor holds x 2^-8: the kernel scales the slabs back
    rc = _C.lib().mi355_linear_partial_img(x.data.data_ptr(), x.M, C.byref(ws_struct), slabs.data_ptr(), max_splits, _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "linear_partial_img")
    return slabs[:rc]


def qkv_rope_kv_write_img(x: ActImage, wqkv: PackedWeight, qkv_bias, cos_sin, positions, block_table, kv_base, scale_base,
                          nh: int, nkv: int, hd: int, page: int, q_len: int = 1, oob_count: Optional[torch.Tensor] = None):
    """qkv_rope_kv_write for 1-64 rows (the step driver: from 5) with the activations as an image; None when the shape is not taken."""
    _chk(x.data, torch.float16, "qkv_rope_kv_write_img.x"); _chk(positions, torch.int32, "positions"); _chk(block_table, torch.int32, "block_table")
    T = x.M
    if x.K != wqkv.K:
        raise _C.Mi355Error(f"qkv_rope_kv_write_img: image K={x.K} against K={wqkv.K}")
    act = _act_of_cache(kv_base) if kv_base.dtype != torch.int8 else torch.float16    # q / bias / cache share the step's 16-bit dtype
    q_out = torch.empty(T, nh, hd, dtype=act, device=x.data.device)
    kv = kv_struct(kv_base, scale_base, page, nkv, hd, act)
    ws_struct = weight_struct(wqkv, act)
    rc = _C.lib().mi355_qkv_rope_kv_write_img(x.data.data_ptr(), T, C.byref(ws_struct), _p(qkv_bias), cos_sin.data_ptr(), hd, cos_sin.shape[0],
                                              positions.data_ptr(), block_table.data_ptr(), block_table.shape[1], q_len, nh,
                                              C.byref(kv), q_out.data_ptr(), _p(oob_count), _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _C.check(rc, "qkv_rope_kv_write_img")
    return q_out


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    _chk_act(x, "rmsnorm.x"); _chk_act(weight, "rmsnorm.weight", x)
    H = x.shape[-1]
    y = torch.empty_like(x)
    _C.check(_C.lib().mi355_rmsnorm_dt(x.data_ptr(), weight.data_ptr(), eps, x.numel() // H, H, y.data_ptr(), _dt(x), _stream()), "rmsnorm")
    return y


def add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float, bias: Optional[torch.Tensor] = None):
    """(y, residual_out): residual_out = x (+bias) + residual;  y = rmsnorm(residual_out) * weight.  fp16 or bf16 throughout."""
    _chk_act(x, "add_rmsnorm.x"); _chk_act(residual, "add_rmsnorm.residual", x); _chk_act(weight, "add_rmsnorm.weight", x)
    if bias is not None:
        _chk_act(bias, "add_rmsnorm.bias", x)
    H = x.shape[-1]
    y, res_out = torch.empty_like(x), torch.empty_like(x)
    _C.check(_C.lib().mi355_add_rmsnorm_dt(x.data_ptr(), None, 0, 0, _p(bias), residual.data_ptr(), res_out.data_ptr(),
                                           weight.data_ptr(), eps, x.numel() // H, H, y.data_ptr(), _dt(x), _stream()), "add_rmsnorm")
    return y, res_out


def silu_mul(gate_up: torch.Tensor) -> torch.Tensor:
    _chk_act(gate_up, "silu_mul.gate_up")
    I = gate_up.shape[-1] // 2
    out = torch.empty(*gate_up.shape[:-1], I, dtype=gate_up.dtype, device=gate_up.device)
    _C.check(_C.lib().mi355_silu_mul_dt(gate_up.data_ptr(), gate_up.numel() // (2 * I), I, out.data_ptr(), _dt(gate_up), _stream()), "silu_mul")
    return out


def embedding(ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    _chk(ids, torch.int32, "embedding.ids"); _chk_act(table, "embedding.table")
    T, (V, H) = ids.numel(), table.shape
    out = torch.empty(T, H, dtype=table.dtype, device=table.device)
    _C.check(_C.lib().mi355_embedding(ids.data_ptr(), T, table.data_ptr(), H, V, out.data_ptr(), _stream()), "embedding")
    return out


def argmax(logits: torch.Tensor) -> torch.Tensor:
    _chk(logits, torch.float32, "argmax.logits")
    B, V = logits.shape
    ld = V
    if V % 4:            # the kernel reads 16-byte vectors: rows of an odd vocabulary go through a row-padded copy
        ld = (V + 3) // 4 * 4
        padded = torch.empty(B, ld, dtype=torch.float32, device=logits.device)
        padded[:, :V] = logits
        logits = padded
    ids = torch.empty(B, dtype=torch.int32, device=logits.device)
    ws = _workspace(B * 64 * 8, logits.device)
    _C.check(_C.lib().mi355_argmax(logits.data_ptr(), B, V, ld, ids.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "argmax")
    return ids


def softmax_rows(logits: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
    """probs = softmax(logits / temperature) per row, fp32 (input of rejection sampling)."""
    _chk(logits, torch.float32, "softmax_rows.logits")
    R, V = logits.shape
    probs = torch.empty_like(logits)
    _C.check(_C.lib().mi355_softmax_rows(logits.data_ptr(), R, V, V, float(temperature), probs.data_ptr(), _stream()), "softmax_rows")
    return probs


def sample_rows(probs: torch.Tensor, uniform: torch.Tensor) -> torch.Tensor:
    """ids[r] ~ probs[r, :] by inverse CDF with uniform[r] in [0, 1) (fp32 prefix sums in index order)."""
    _chk(probs, torch.float32, "sample_rows.probs"); _chk(uniform, torch.float32, "sample_rows.uniform")
    R, V = probs.shape
    if uniform.numel() != R:
        raise _C.Mi355Error("sample_rows: one uniform per row")
    ids = torch.empty(R, dtype=torch.int32, device=probs.device)
    _C.check(_C.lib().mi355_sample_rows(probs.data_ptr(), R, V, V, uniform.data_ptr(), ids.data_ptr(), _stream()), "sample_rows")
    return ids


def apply_penalties(logits: torch.Tensor, temperature: Optional[torch.Tensor] = None, repetition_penalty: Optional[torch.Tensor] = None,
                    presence_penalty: Optional[torch.Tensor] = None, frequency_penalty: Optional[torch.Tensor] = None,
                    output_ids: Optional[torch.Tensor] = None, input_lengths: Optional[torch.Tensor] = None,
                    max_input_length: int = 0, step: int = 0) -> torch.Tensor:
    """In place on fp32 logits [B, V]: temperature (logit / (T + 1e-6)), then repetition / presence / frequency penalties over
    the token history output_ids [step, B] int32 (bindings/common/kernels/sampling_penalty_kernels.cu:26-54,129-213)."""
    _chk(logits, torch.float32, "apply_penalties.logits")
    B, V = logits.shape
    dev = logits.device

    def vec(t, dt, name):
        if t is None:
            return None
        t = t.to(device=dev, dtype=dt).contiguous()
        if t.numel() != B:
            raise _C.Mi355Error(f"apply_penalties: {name} needs one value per row")
        return t
    temp, rep = vec(temperature, torch.float32, "temperature"), vec(repetition_penalty, torch.float32, "repetition_penalty")
    pres, freq = vec(presence_penalty, torch.float32, "presence_penalty"), vec(frequency_penalty, torch.float32, "frequency_penalty")
    lens = vec(input_lengths, torch.int32, "input_lengths")
    ws = None
    if rep is not None or pres is not None or freq is not None:
        if output_ids is None:
            raise _C.Mi355Error("apply_penalties: penalties need the token history output_ids [step, B]")
        _chk(output_ids, torch.int32, "apply_penalties.output_ids")
        if output_ids.dim() != 2 or output_ids.shape[1] != B or output_ids.shape[0] < step:
            raise _C.Mi355Error("apply_penalties: output_ids must be [>= step, B]")
        ws = torch.empty(B, V, dtype=torch.int32, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None
    _C.check(_C.lib().mi355_apply_penalties(logits.data_ptr(), B, V, V, ptr(temp), ptr(rep), ptr(pres), ptr(freq), ptr(output_ids),
                                            ptr(lens), int(max_input_length), int(step), ptr(ws), _stream()), "apply_penalties")
    return logits


def ban_repeat_ngram(logits: torch.Tensor, token_ids: torch.Tensor, sequence_last_index: torch.Tensor,
                     no_repeat_ngram_size: torch.Tensor) -> torch.Tensor:
    """In place on fp32 logits [B', V] (the first B rows are touched): token_ids [B, L] int32, sequence_last_index [B],
    no_repeat_ngram_size [B] (bindings/common/kernels/banRepeatNgram.cu:30-136)."""
    _chk(logits, torch.float32, "ban_repeat_ngram.logits"); _chk(token_ids, torch.int32, "ban_repeat_ngram.token_ids")
    dev = logits.device
    B, L = token_ids.shape
    last = sequence_last_index.to(device=dev, dtype=torch.int32).contiguous()
    ng = no_repeat_ngram_size.to(device=dev, dtype=torch.int32).contiguous()
    if last.numel() != B or ng.numel() != B or B > logits.shape[0]:
        raise _C.Mi355Error("ban_repeat_ngram: one last index / n-gram size per history row, at most one row per logit row")
    _C.check(_C.lib().mi355_ban_repeat_ngram(logits.data_ptr(), B, logits.shape[1], logits.shape[1], token_ids.data_ptr(), L,
                                             last.data_ptr(), ng.data_ptr(), _stream()), "ban_repeat_ngram")
    return logits


def top_k_top_p_sample(probs: torch.Tensor, top_k: Optional[torch.Tensor], top_p: Optional[torch.Tensor], uniform: torch.Tensor,
                       return_probs: bool = False):
    """Per row of fp32 probabilities: top-k then top-p filter, renormalise, draw by inverse CDF with uniform[r]
    (bindings/core/CudaSampleOp.cc:748-786).  -> ids [R] int32 (, renormalised filtered probs [R, V])."""
    _chk(probs, torch.float32, "top_k_top_p_sample.probs"); _chk(uniform, torch.float32, "top_k_top_p_sample.uniform")
    R, V = probs.shape
    dev = probs.device
    if uniform.numel() != R:
        raise _C.Mi355Error("top_k_top_p_sample: one uniform per row")
    k = top_k.to(device=dev, dtype=torch.int32).contiguous() if top_k is not None else None
    p = top_p.to(device=dev, dtype=torch.float32).contiguous() if top_p is not None else None
    if (k is not None and k.numel() != R) or (p is not None and p.numel() != R):
        raise _C.Mi355Error("top_k_top_p_sample: one top_k / top_p per row")
    ids = torch.empty(R, dtype=torch.int32, device=dev)
    out = torch.empty_like(probs) if return_probs else None
    _C.check(_C.lib().mi355_top_k_top_p_sample(probs.data_ptr(), R, V, V, k.data_ptr() if k is not None else None,
                                               p.data_ptr() if p is not None else None, uniform.data_ptr(), ids.data_ptr(),
                                               out.data_ptr() if out is not None else None, V, _stream()), "top_k_top_p_sample")
    return (ids, out) if return_probs else ids


def rejection_sample(draft_token_ids: torch.Tensor, target_token_ids: torch.Tensor, target_probs: torch.Tensor,
                     uniform_samples: torch.Tensor, do_sample: torch.Tensor, draft_probs: Optional[torch.Tensor] = None):
    """Chain rejection sampling (bindings/rocm/speculative_sampling/sampling.cu:306): draft_token_ids [B,g] int32,
    target_token_ids [B,g+1] or [B,g+1,stride] int32, target_probs [B,g+1,V] fp32, uniform_samples [B,g+1] fp32,
    do_sample [B] bool/uint8, draft_probs [B,g,V] fp32 or None (point mass).  -> (output_token_ids [B,g+1], accepted [B])."""
    _chk(draft_token_ids, torch.int32, "rejection_sample.draft_token_ids")
    _chk(target_token_ids, torch.int32, "rejection_sample.target_token_ids")
    _chk(target_probs, torch.float32, "rejection_sample.target_probs")
    _chk(uniform_samples, torch.float32, "rejection_sample.uniform_samples")
    B, G = draft_token_ids.shape
    V = target_probs.shape[-1]
    if target_probs.shape != (B, G + 1, V) or uniform_samples.shape != (B, G + 1) or target_token_ids.shape[:2] != (B, G + 1):
        raise _C.Mi355Error("rejection_sample: inconsistent shapes")
    stride = 1 if target_token_ids.dim() == 2 else target_token_ids.shape[2]
    ds = do_sample.to(torch.uint8).contiguous()
    if draft_probs is not None:
        _chk(draft_probs, torch.float32, "rejection_sample.draft_probs")
        if draft_probs.shape != (B, G, V):
            raise _C.Mi355Error("rejection_sample: draft_probs shape")
    out = torch.empty(B, G + 1, dtype=torch.int32, device=draft_token_ids.device)
    acc = torch.empty(B, dtype=torch.int32, device=draft_token_ids.device)
    _C.check(_C.lib().mi355_rejection_sample(_p(draft_probs), draft_token_ids.data_ptr(), uniform_samples.data_ptr(),
                                             target_probs.data_ptr(), target_token_ids.data_ptr(), stride, out.data_ptr(),
                                             acc.data_ptr(), ds.data_ptr(), B, G, V, 1 if draft_probs is None else 0, _stream()),
             "rejection_sample")
    return out, acc


# ------------------------------------------------------------------ attention
def rope_kv_write(qkv: torch.Tensor, qkv_bias: Optional[torch.Tensor], cos_sin: torch.Tensor, positions: torch.Tensor,
                  block_table: torch.Tensor, kv_base: torch.Tensor, scale_base: Optional[torch.Tensor], nh: int, nkv: int,
                  hd: int, page: int, oob_count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RoPE + bias + Q-extract + paged KV write for decode tokens; returns q [T, nh, hd].  Tokens with a position or
    block id out of range are not written; `oob_count` (int32 [1], device) counts them."""
    _chk_act(qkv, "rope_kv_write.qkv"); _chk(cos_sin, torch.float32, "rope_kv_write.cos_sin")
    _chk(positions, torch.int32, "rope_kv_write.positions"); _chk(block_table, torch.int32, "rope_kv_write.block_table")
    if qkv_bias is not None:
        _chk(qkv_bias, qkv.dtype, "rope_kv_write.qkv_bias")
    T = qkv.shape[0]
    q_out = torch.empty(T, nh, hd, dtype=qkv.dtype, device=qkv.device)
    kv = kv_struct(kv_base, scale_base, page, nkv, hd, qkv.dtype)
    _C.check(_C.lib().mi355_rope_kv_write(qkv.data_ptr(), None, 0, qkv.shape[1], _p(qkv_bias), cos_sin.data_ptr(), hd,
                                          cos_sin.shape[0], positions.data_ptr(), block_table.data_ptr(),
                                          block_table.shape[1], T, nh, C.byref(kv), q_out.data_ptr(), _p(oob_count), _stream()),
             "rope_kv_write")
    return q_out


def rope_kv_write_rows(qkv: torch.Tensor, qkv_bias, cos_sin, positions, block_table, kv_base, scale_base, nh: int, nkv: int,
                       hd: int, page: int, q_len: int, oob_count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q_len rows per sequence: qkv [B*q_len, ...], positions [B*q_len] (< 0 = padding row), block_table [B, M]."""
    _chk_act(qkv, "rope_kv_write_rows.qkv"); _chk(positions, torch.int32, "positions"); _chk(block_table, torch.int32, "block_table")
    if qkv_bias is not None:
        _chk(qkv_bias, qkv.dtype, "rope_kv_write_rows.qkv_bias")
    T = qkv.shape[0]
    q_out = torch.empty(T, nh, hd, dtype=qkv.dtype, device=qkv.device)
    kv = kv_struct(kv_base, scale_base, page, nkv, hd, qkv.dtype)
    _C.check(_C.lib().mi355_rope_kv_write_rows(qkv.data_ptr(), None, 0, qkv.shape[1], _p(qkv_bias), cos_sin.data_ptr(), hd,
                                               cos_sin.shape[0], positions.data_ptr(), block_table.data_ptr(), block_table.shape[1],
                                               T, q_len, nh, C.byref(kv), q_out.data_ptr(), _p(oob_count), _stream()),
             "rope_kv_write_rows")
    return q_out


def paged_attention_rows(q: torch.Tensor, kv_base, scale_base, block_table: torch.Tensor, positions: torch.Tensor, nkv: int,
                         page: int, q_len: int, max_seq_len: int, scale: Optional[float] = None) -> torch.Tensor:
    """q [B*q_len, nh, hd]; row i of sequence b attends tokens 0..positions[b*q_len+i] (causal over the paged cache)."""
    _chk_act(q, "paged_attention_rows.q"); _chk(block_table, torch.int32, "block_table"); _chk(positions, torch.int32, "positions")
    T, nh, hd = q.shape
    kv = kv_struct(kv_base, scale_base, page, nkv, hd, q.dtype)
    out = torch.empty(T, nh * hd, dtype=q.dtype, device=q.device)
    need = _C.lib().mi355_paged_attn_workspace_bytes(T, nh, hd, max_seq_len)
    ws = _workspace(need, q.device)
    _C.check(_C.lib().mi355_paged_attn_rows(q.data_ptr(), C.byref(kv), block_table.data_ptr(), block_table.shape[1], positions.data_ptr(),
                                            T // q_len, q_len, nh, scale if scale is not None else 1.0 / math.sqrt(hd), max_seq_len,
                                            out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "paged_attn_rows")
    return out


def paged_decode_attention(q: torch.Tensor, kv_base: torch.Tensor, scale_base: Optional[torch.Tensor],
                           block_table: torch.Tensor, seq_lens: torch.Tensor, nkv: int, page: int,
                           max_seq_len: int, scale: Optional[float] = None) -> torch.Tensor:
    """q [B, nh, hd] -> out [B, nh*hd]; seq_lens = context length including the new token."""
    _chk_act(q, "paged_decode_attention.q"); _chk(block_table, torch.int32, "block_table")
    _chk(seq_lens, torch.int32, "seq_lens")
    B, nh, hd = q.shape
    kv = kv_struct(kv_base, scale_base, page, nkv, hd, q.dtype)
    out = torch.empty(B, nh * hd, dtype=q.dtype, device=q.device)
    need = _C.lib().mi355_paged_attn_workspace_bytes(B, nh, hd, max_seq_len)
    ws = _workspace(need, q.device)
    _C.check(_C.lib().mi355_paged_decode_attn(q.data_ptr(), C.byref(kv), block_table.data_ptr(), block_table.shape[1],
                                              seq_lens.data_ptr(), B, nh, scale if scale is not None else 1.0 / math.sqrt(hd),
                                              max_seq_len, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
             "paged_decode_attn")
    return out
