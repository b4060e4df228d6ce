"""Decode attention strategy — the reference's FMHA plug-in surface
(rtp_llm/models_py/modules/factory/attention/{fmha_impl_base.py:99-178, attn_factory.py:100-254}):

    DECODE_MHA_IMPS.append(cls); cls.support(attn_configs, attn_inputs);
    impl = cls(attn_configs, attn_inputs, parallelism_config); impl.forward(qkv, LayerKVCache, layer_idx)

Data-contract mirrors (bindings/OpDefs.h:29-51,281-327; cpp/model_utils/AttentionConfig.h:24-85):
LayerKVCache, PyAttentionInputs, AttentionConfigs — same field names, decode fields only.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import ops


@dataclass
class AttentionConfigs:       # cpp/model_utils/AttentionConfig.h:24-85 (per-rank values)
    head_num: int
    kv_head_num: int
    size_per_head: int
    rope_dim: int = 0
    rope_theta: float = 1e6
    max_seq_len: int = 4096
    softmax_extra_scale: float = 1.0
    kernel_tokens_per_block: int = 16


@dataclass
class LayerKVCache:           # bindings/OpDefs.h:29-51
    kv_cache_base: torch.Tensor                    # [blocks, 2, nkv, page, hd]
    kv_scale_base: Optional[torch.Tensor] = None   # [blocks, 2*nkv*page] fp32 (8-bit caches)
    seq_size_per_block: int = 0
    layer_id: int = -1


@dataclass
class PyAttentionInputs:      # bindings/OpDefs.h:281-327, decode fields
    is_prefill: bool = False
    sequence_lengths: torch.Tensor = None              # int32 [B]: tokens already in the cache
    input_lengths: torch.Tensor = None
    kv_cache_kernel_block_id_device: torch.Tensor = None   # int32 [B, M]
    is_cuda_graph: bool = False
    sequence_lengths_plus_1_device: torch.Tensor = None    # filled by prepare()


class FMHAImplBase:
    def forward(self, qkv, kv_cache, layer_idx: int = 0):
        raise NotImplementedError

    @staticmethod
    def support(attn_configs, attn_inputs) -> bool:
        return False

    def support_cuda_graph(self) -> bool:       # fmha_impl_base.py:165-172
        return callable(getattr(self, "prepare_cuda_graph", None))


class Mi355PagedDecodeImpl(FMHAImplBase):
    """RoPE+KV-write op followed by the paged flash-decoding op (the pairing of
    AiterDecodeImpl*, rocm_impl/aiter.py:1960-2031)."""

    def __init__(self, attn_configs: AttentionConfigs, attn_inputs: PyAttentionInputs, parallelism_config=None,
                 cos_sin: Optional[torch.Tensor] = None, qkv_bias: Optional[torch.Tensor] = None):
        self.cfg = attn_configs
        self.cos_sin = cos_sin
        self.prepare(attn_inputs)

    @staticmethod
    def support(attn_configs: AttentionConfigs, attn_inputs: PyAttentionInputs) -> bool:
        g = attn_configs.head_num // max(1, attn_configs.kv_head_num)
        return (not attn_inputs.is_prefill and attn_configs.size_per_head in (64, 128) and g <= 16
                and attn_configs.head_num % attn_configs.kv_head_num == 0
                and attn_configs.rope_dim in (0, attn_configs.size_per_head))

    def prepare(self, attn_inputs: PyAttentionInputs):
        """Host-side prep (FMHAParams, aiter.py:205-209: seq_lens = sequence_lengths + 1 on device)."""
        self.inputs = attn_inputs
        dev = attn_inputs.kv_cache_kernel_block_id_device.device
        self.positions = attn_inputs.sequence_lengths.to(device=dev, dtype=torch.int32).contiguous()
        self.seq_lens = (self.positions + 1).contiguous()
        attn_inputs.sequence_lengths_plus_1_device = self.seq_lens
        self.block_table = attn_inputs.kv_cache_kernel_block_id_device.to(torch.int32).contiguous()

    def prepare_cuda_graph(self, attn_inputs: PyAttentionInputs):
        """In-place refresh of the address-stable device buffers for graph replay (aiter.py:1964-1981)."""
        self.positions.copy_(attn_inputs.sequence_lengths.to(self.positions.device, torch.int32))
        torch.add(self.positions, 1, out=self.seq_lens)
        self.block_table.copy_(attn_inputs.kv_cache_kernel_block_id_device)

    def forward(self, qkv: torch.Tensor, kv_cache: LayerKVCache, layer_idx: int = 0) -> torch.Tensor:
        c = self.cfg
        page = kv_cache.seq_size_per_block or kv_cache.kv_cache_base.shape[3]
        q = ops.rope_kv_write(qkv.contiguous(), None, self.cos_sin, self.positions, self.block_table, kv_cache.kv_cache_base,
                              kv_cache.kv_scale_base, c.head_num, c.kv_head_num, c.size_per_head, page)
        return ops.paged_decode_attention(q, kv_cache.kv_cache_base, kv_cache.kv_scale_base, self.block_table, self.seq_lens,
                                          c.kv_head_num, page, c.max_seq_len,
                                          c.softmax_extra_scale / math.sqrt(c.size_per_head))


DECODE_MHA_IMPS: List[type] = [Mi355PagedDecodeImpl]   # attention/__init__.py:47-49 registration list


class AttnImplFactory:
    """attn_factory.py:226-254: first registered impl whose support() accepts the inputs."""

    @staticmethod
    def get_fmha_impl(attn_configs, attn_inputs, parallelism_config=None, **kw) -> FMHAImplBase:
        for impl in DECODE_MHA_IMPS:
            if impl.support(attn_configs, attn_inputs):
                inst = impl(attn_configs, attn_inputs, parallelism_config, **kw)
                if attn_inputs.is_cuda_graph and not inst.support_cuda_graph():
                    continue
                return inst
        raise RuntimeError("no decode FMHA implementation supports these inputs")
