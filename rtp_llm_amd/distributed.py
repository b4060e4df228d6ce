"""TP collectives with the reference's call shape (rtp_llm/models_py/distributed/collective_torch.py:694-769):
all_reduce(tensor, Group.TP) / all_gather(tensor, Group.TP).  Backend: torch.distributed — "nccl" is RCCL
over xGMI on ROCm, "gloo" for the CPU tests.  One process per GPU."""
import enum
import os
from typing import Optional

import torch
import torch.distributed as dist


class Group(enum.Enum):
    DP = 0
    TP = 1
    DP_AND_TP = 2


_tp_group = None


def init_distributed(backend: Optional[str] = None) -> None:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend)


def set_tp_group(group) -> None:
    global _tp_group
    _tp_group = group


def tp_group():
    return _tp_group


def tp_size() -> int:
    if not dist.is_initialized():
        return 1
    return dist.get_world_size(_tp_group)


def tp_rank() -> int:
    if not dist.is_initialized():
        return 0
    return dist.get_rank(_tp_group)


def all_reduce(tensor: torch.Tensor, group: Group = Group.TP, inplace: bool = True) -> torch.Tensor:
    """SUM over the TP ranks (C1/C2 of SURVEY 2.3: after O-proj and down-proj)."""
    if tp_size() == 1:
        return tensor
    t = tensor if inplace else tensor.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_tp_group)
    return t


def all_gather(tensor: torch.Tensor, group: Group = Group.TP) -> torch.Tensor:
    """Concatenate along the last dim over TP ranks (logits gather, PyWrappedModel.cc:915-936)."""
    n = tp_size()
    if n == 1:
        return tensor
    outs = [torch.empty_like(tensor) for _ in range(n)]
    dist.all_gather(outs, tensor.contiguous(), group=_tp_group)
    return torch.cat(outs, dim=-1)
