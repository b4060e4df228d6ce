"""Linear strategies — same plug-in surface as the reference's LinearFactory
(rtp_llm/models_py/modules/factory/linear/{linear_base.py:16-102, factory.py:33-145}):

    LinearFactory.register(cls); cls.can_handle(...) must be unique among strategies;
    cls(weight, weight_scales, input_scales, bias, quant_config, weight_scale_2); forward(x[M,K]) -> [M,N]

The reference has no W4A16/W8A16 strategy in this snapshot (an int8/GPTQ/AWQ weight
dict matches no strategy and raises); these classes fill that slot on MI355X, plus an fp16
strategy for unquantised layers (lm_head).  One extension over the reference factory: the
group-wise zeros (``W.*_z``) must reach the strategy — ``create_linear_from_weights`` forwards
``zeros_key`` (the reference drops it, factory.py:86-93; SURVEY 8b "Gap").
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Dict, List, Optional, Type

import torch
from torch import nn

from . import _C, ops, quant
from .quant import PackedWeight


@dataclass
class QuantConfig:
    """Subset of rtp_llm/config/quant_config.py:281-300,641-685 that the decode path reads."""
    method: str = "none"        # "none" | "int8" (load-time autoquant) | "gptq" | "awq"
    bits: int = 16
    group_size: int = 0

    def is_quanted(self) -> bool:
        return self.method != "none"


class LinearBase(nn.Module, ABC):
    """Same abstract surface as the reference's LinearBase (linear_base.py:16-102)."""

    @classmethod
    @abstractmethod
    def can_handle(cls, quant_config, weight, weight_scales, hw_kernel_config=None, weight_scale_2=None,
                   input_scale=None) -> bool:
        ...

    @abstractmethod
    def __init__(self, weight, weight_scales=None, input_scales=None, bias=None, quant_config=None, weight_scale_2=None):
        super().__init__()

    @abstractmethod
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        ...

    def maybe_cache_quant_scale(self, max_len: int) -> None:  # linear_base.py:66-73 (no-op for weight-only)
        pass

    def __repr__(self) -> str:
        return self.__class__.__name__


# ---- strategy implementations: plain mixins, so the same code binds to this package's LinearBase (below) and to the
# reference's own LinearBase (rtp_llm_amd/reference_plugin.py) without a second copy
class _PackedImpl:
    """Common forward: one C-ABI call on the packed weight (bias fused in the GEMM epilogue, as the reference applies
    Qwen2's QKV bias inside the linear, causal_attention.py:42-51)."""
    packed: PackedWeight
    kernel_hints: int = 0   # _C.HINT_* bits forwarded with every call (A/B tests); 0 = let the library choose

    def _finish_init(self, packed: PackedWeight, bias: Optional[torch.Tensor]):
        self.packed = packed
        self.bias = None if bias is None else bias.to(torch.float16).contiguous()
        self.in_features, self.out_features = packed.K, packed.N

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return ops.linear(input.contiguous(), self.packed, self.bias, epilogue=self.kernel_hints)

    def forward_silu_mul(self, input: torch.Tensor) -> torch.Tensor:
        """gate_up projection + SiLU-gate in one kernel; requires interleaved (gate, up) columns."""
        assert getattr(self, "gate_up_interleaved", False), "weight was not packed with interleaved gate/up columns"
        return ops.linear(input.contiguous(), self.packed, self.bias, epilogue=_C.EPI_SILU_MUL | self.kernel_hints)


class _F16Impl(_PackedImpl):
    """fp16 weights [K, N] (the reference stores [K, N] and calls hipb_mm, f16_linear.py:100-112)."""

    @classmethod
    def can_handle(cls, quant_config, weight, weight_scales, hw_kernel_config=None, weight_scale_2=None, input_scale=None):
        return weight_scales is None and weight.dtype == torch.float16

    def __init__(self, weight, weight_scales=None, input_scales=None, bias=None, quant_config=None, weight_scale_2=None):
        super().__init__(weight, weight_scales, input_scales, bias, quant_config, weight_scale_2)
        self._finish_init(quant.pack_fp16(weight.contiguous()), bias)


class _W8A16Impl(_PackedImpl):
    """INT8 weight-only, per-output-channel scale (load-time autoquant, device_impl.py:183-222).
    weight int8 [K, N], weight_scales [N]."""

    @classmethod
    def can_handle(cls, quant_config, weight, weight_scales, hw_kernel_config=None, weight_scale_2=None, input_scale=None):
        return weight_scales is not None and weight.dtype == torch.int8 and weight_scales.dim() == 1

    def __init__(self, weight, weight_scales=None, input_scales=None, bias=None, quant_config=None, weight_scale_2=None):
        super().__init__(weight, weight_scales, input_scales, bias, quant_config, weight_scale_2)
        self._finish_init(quant.pack_int8_per_channel(weight.contiguous(), weight_scales), bias)


def _is_reference_w4_kernel(weight, weight_scales) -> bool:
    """int8 [K/2, N] + scales [K/g, N], g in {32, 64, 128}: the tensors RocmImpl.preprocess_groupwise_weight_params emits."""
    if weight_scales is None or weight.dtype != torch.int8 or weight_scales.dim() != 2 or weight.dim() != 2:
        return False
    K, G = 2 * weight.shape[0], weight_scales.shape[0]
    return weight.shape[1] == weight_scales.shape[1] and G > 0 and K % G == 0 and K // G in (32, 64, 128)


class _W4A16Impl(_PackedImpl):
    """INT4 group-wise (GPTQ / AWQ).  Two input forms:
      * canonical: weight uint8 codes [K, N] in 0..15, weight_scales fp16 [K/g, N], weight_zeros = effective zero codes
        [K/g, N] (z + 1 for GPTQ);
      * what the reference's ROCm loader hands its factory (device_impl.py:797-868): weight int8 [K/2, N] CK-permuted
        nibble pairs, weight_scales [K/g, N], and zeros_x_scales [K/g, N] fp16.  The reference factory never forwards the
        zeros (factory.py:86-93; SURVEY 8b "Gap"), so they are taken from, in this order: the `weight_zeros` keyword (the
        small factory extension of INTEGRATION.md), the `weight_scale_2` slot (forwarded by the UNMODIFIED reference
        factory: pass weight_scale_2_key=W.*_z at the call site), or a `zeros_x_scales` attribute set on the weight tensor."""

    @classmethod
    def can_handle(cls, quant_config, weight, weight_scales, hw_kernel_config=None, weight_scale_2=None, input_scale=None):
        if weight_scales is None or weight_scales.dim() != 2:
            return False
        return weight.dtype == torch.uint8 or _is_reference_w4_kernel(weight, weight_scales)

    def __init__(self, weight, weight_scales=None, input_scales=None, bias=None, quant_config=None, weight_scale_2=None,
                 weight_zeros=None):
        super().__init__(weight, weight_scales, input_scales, bias, quant_config, weight_scale_2)
        if weight.dtype == torch.int8:     # reference loader format
            zs = weight_zeros if weight_zeros is not None else weight_scale_2 if weight_scale_2 is not None \
                else getattr(weight, "zeros_x_scales", None)
            if zs is None or tuple(zs.shape) != tuple(weight_scales.shape):
                raise ValueError("W4A16: zeros_x_scales [K/g, N] is required with the reference's packed int4 kernel "
                                 "(weight_zeros=, the weight_scale_2 slot, or weight.zeros_x_scales)")
            self._finish_init(quant.pack_reference_rocm_w4(weight, weight_scales, zs), bias)
            return
        gs = weight.shape[0] // weight_scales.shape[0]
        if weight_zeros is None:  # symmetric: stored code = q_s + 8
            weight_zeros = torch.full_like(weight_scales, 8, dtype=torch.int16)
        self._finish_init(quant.pack_groupwise_w4(weight, weight_zeros, weight_scales, gs), bias)


class Mi355F16Linear(_F16Impl, LinearBase):
    pass


class Mi355W8A16Linear(_W8A16Impl, LinearBase):
    pass


class Mi355W4A16Linear(_W4A16Impl, LinearBase):
    pass


STRATEGY_IMPLS = (("Mi355F16Linear", _F16Impl), ("Mi355W8A16Linear", _W8A16Impl), ("Mi355W4A16Linear", _W4A16Impl))


class LinearFactory:
    """factory.py:33-145 — registry + unique-match dispatch."""
    _strategies: List[Type[LinearBase]] = []

    @classmethod
    def register(cls, strategy: Type[LinearBase]) -> None:
        if strategy not in cls._strategies:
            cls._strategies.append(strategy)

    @classmethod
    def create_linear(cls, weight, bias=None, weight_scales=None, quant_config=None, weight_zeros=None, **kw) -> LinearBase:
        hits = [s for s in cls._strategies if s.can_handle(quant_config, weight, weight_scales)]
        if len(hits) != 1:  # factory.py:106-127: zero or several matches is an error
            raise ValueError(f"expected exactly one Linear strategy, got {[h.__name__ for h in hits]} for "
                             f"weight dtype {weight.dtype}, scales {None if weight_scales is None else tuple(weight_scales.shape)}")
        extra = {"weight_zeros": weight_zeros} if weight_zeros is not None else {}
        return hits[0](weight, weight_scales, None, bias, quant_config, None, **extra)

    @classmethod
    def create_linear_from_weights(cls, weights: Dict[str, torch.Tensor], weight_key: str, scale_key: Optional[str] = None,
                                   bias_key: Optional[str] = None, quant_config=None, zeros_key: Optional[str] = None) -> LinearBase:
        return cls.create_linear(weights[weight_key], weights.get(bias_key) if bias_key else None,
                                 weights.get(scale_key) if scale_key else None, quant_config,
                                 weights.get(zeros_key) if zeros_key else None)


for _s in (Mi355F16Linear, Mi355W8A16Linear, Mi355W4A16Linear):
    LinearFactory.register(_s)
