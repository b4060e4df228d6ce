"""Bind the MI355 linear strategies to the REFERENCE's own plug-in classes.

The reference picks a linear implementation by asking every class registered with its ``LinearFactory`` whether it
``can_handle`` the weight dict (rtp_llm/models_py/modules/factory/linear/factory.py:33-145; registration at import time
in impl/rocm/__init__.py:16-20).  ``register(LinearFactory, LinearBase)`` creates subclasses of the reference's
``LinearBase`` (linear_base.py:16-102) around this package's strategy implementations and registers them, which fills the
W4A16 / W8A16 slot the snapshot leaves empty -- model code such as ``CausalAttention`` / ``DenseMLP`` then
runs unchanged.  The reference-side call is one line in impl/rocm/__init__.py.
"""
from typing import List

from .linear import STRATEGY_IMPLS


def make_strategies(ref_linear_base) -> List[type]:
    """Subclasses of the reference's LinearBase, one per MI355 strategy."""
    return [type(name, (impl, ref_linear_base), {"__module__": __name__, "__doc__": impl.__doc__}) for name, impl in STRATEGY_IMPLS]


def register(ref_linear_factory, ref_linear_base, replace_f16: bool = False) -> List[type]:
    """Register the strategies with the reference's factory.  The fp16 strategy overlaps the reference's own ROCm fp16
    strategies (RocmF16Linear*), and the factory demands a unique match, so it is only registered when asked to replace
    them (``replace_f16``: existing fp16 strategies are removed first)."""
    out = []
    for cls in make_strategies(ref_linear_base):
        if cls.__name__ == "Mi355F16Linear":
            if not replace_f16:
                continue
            import torch
            probe = torch.zeros(8, 8, dtype=torch.float16)
            ref_linear_factory._strategies[:] = [s for s in ref_linear_factory._strategies
                                                 if not _safe_can_handle(s, probe)]
        ref_linear_factory.register(cls)
        out.append(cls)
    return out


def _safe_can_handle(strategy, weight) -> bool:
    try:
        return bool(strategy.can_handle(None, weight, None, None, None, None))
    except Exception:  # noqa: BLE001  (a strategy that cannot even inspect an fp16 weight does not claim it)
        return False
