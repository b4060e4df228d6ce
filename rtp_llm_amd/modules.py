"""Leaf modules with the reference's call shapes (rtp_llm/models_py/modules/base/__init__.py:26-60):
RMSNorm(weight, eps)(x), RMSResNorm(weight, eps)(x, residual) -> (y, residual_out),
FusedSiluAndMul()(gate_up), Embedding(weight)(ids).  Each forwards to one HIP kernel."""
import torch
from torch import nn

from . import ops


class RMSNorm(nn.Module):
    """modules/base/rocm/norm.py:51-56 (aiter.rms_norm) -> ops.rmsnorm."""

    def __init__(self, weight: torch.Tensor, eps: float = 1e-6):
        super().__init__()
        self.weight, self.eps = weight, eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.rmsnorm(x.contiguous(), self.weight, self.eps)


class RMSResNorm(nn.Module):
    """modules/base/rocm/norm.py:59-77 (aiter.rmsnorm2d_fwd_with_add) -> ops.add_rmsnorm."""

    def __init__(self, weight: torch.Tensor, eps: float = 1e-6):
        super().__init__()
        self.weight, self.eps = weight, eps

    def forward(self, x: torch.Tensor, residual: torch.Tensor):
        return ops.add_rmsnorm(x.contiguous(), residual.contiguous(), self.weight, self.eps)


class FusedSiluAndMul(nn.Module):
    """modules/base/rocm/activation.py:9-24 (aiter.silu_and_mul) -> ops.silu_mul."""

    def forward(self, gate_up: torch.Tensor) -> torch.Tensor:
        return ops.silu_mul(gate_up.contiguous())


class Embedding(nn.Module):
    """modules/base/common/embedding.py:22-59 (rtp_llm_ops.embedding); TP all-gather variant lives in
    rtp_llm_amd.distributed (hidden-split table, :50-58)."""

    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.weight = weight

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        return ops.embedding(ids.to(torch.int32).contiguous(), self.weight)
