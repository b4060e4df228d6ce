"""Paged KV-cache layout helpers (host side, torch ops, device-agnostic).

The allocator contract is the reference's: a per-layer tensor
``[blocks, 2, nkv, page, hd]`` (bindings/OpDefs.h:24-28,201-202), K block = index 0,
V block = index 1 of dim 1 (pool index 2b / 2b+1, kv_cache_kernels.cu:58-61), plus an
fp32 scale plane ``[blocks, 2, nkv, page]`` for 8-bit caches (MHAKVCacheSpec.h:52-54).
Inside a block the layout is MI355-native (we own the writer and every reader):

    K block: [nkv][page][hd]     V block: [nkv][hd][page]

8-bit caches hold the codes offset-binary (code + 128, i.e. the int8 with its top bit flipped): the attention kernel widens
bytes by placing them under an fp16 exponent, which wants unsigned bytes.  write_tokens / read_tokens take and return the
signed codes.

The hot-path writer is the HIP kernel behind ``ops.rope_kv_write``; the functions here
are the slow-path equivalents used to import a prefilled cache (the prefill path itself
is out of scope this round) and by the tests to read the cache back.
"""
from typing import Optional, Tuple

import torch


def alloc_layer_cache(num_blocks: int, nkv: int, page: int, hd: int, int8: bool, device,
                      dtype: torch.dtype = torch.float16) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """dtype: element type of a 16-bit cache (fp16 or bf16 -- the activation dtype of the model); ignored for INT8."""
    dt = torch.int8 if int8 else dtype
    kv = torch.zeros(num_blocks, 2, nkv, page, hd, dtype=dt, device=device)
    sc = torch.ones(num_blocks, 2, nkv, page, dtype=torch.float32, device=device) if int8 else None
    return kv, sc


def _flip(x: torch.Tensor) -> torch.Tensor:
    """signed code <-> stored byte (top bit flipped)."""
    return torch.bitwise_xor(x, torch.tensor(-128, dtype=torch.int8, device=x.device))


def _views(kv_base: torch.Tensor):
    nb, _, nkv, page, hd = kv_base.shape
    flat = kv_base.view(nb, 2, nkv, page * hd)
    k = flat[:, 0].unflatten(-1, (page, hd))   # [nb, nkv, page, hd]
    v = flat[:, 1].unflatten(-1, (hd, page))   # [nb, nkv, hd, page]
    return k, v


def write_tokens(kv_base: torch.Tensor, scale_base: Optional[torch.Tensor], block_table_row: torch.Tensor, start: int,
                 K: torch.Tensor, V: torch.Tensor, k_scale: Optional[torch.Tensor] = None,
                 v_scale: Optional[torch.Tensor] = None) -> None:
    """Store tokens start..start+T-1 of one sequence.  K, V: [T, nkv, hd] in the cache dtype;
    k_scale, v_scale: [T, nkv] fp32 for an int8 cache."""
    page = kv_base.shape[3]
    T = K.shape[0]
    pos = torch.arange(start, start + T, device=kv_base.device)
    blk = block_table_row.to(kv_base.device).long()[pos // page]
    off = pos % page
    kview, vview = _views(kv_base)
    K, V = K.to(kv_base.device), V.to(kv_base.device)
    if kv_base.dtype == torch.int8:
        K, V = _flip(K), _flip(V)
    kview[blk, :, off, :] = K
    vview[blk, :, :, off] = V
    if scale_base is not None:
        scale_base[blk, 0, :, off] = k_scale.to(kv_base.device)
        scale_base[blk, 1, :, off] = v_scale.to(kv_base.device)


def read_tokens(kv_base: torch.Tensor, scale_base: Optional[torch.Tensor], block_table_row: torch.Tensor, ctx: int):
    """Natural view of the first ctx tokens of one sequence: (K, V [ctx,nkv,hd], k_scale, v_scale [ctx,nkv] or None)."""
    page = kv_base.shape[3]
    pos = torch.arange(ctx, device=kv_base.device)
    blk = block_table_row.to(kv_base.device).long()[pos // page]
    off = pos % page
    kview, vview = _views(kv_base)
    K, V = kview[blk, :, off, :], vview[blk, :, :, off]
    if scale_base is None:
        return K, V, None, None
    return _flip(K), _flip(V), scale_base[blk, 0, :, off], scale_base[blk, 1, :, off]
