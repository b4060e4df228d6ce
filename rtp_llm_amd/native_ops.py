"""Loader of the native registration shim (lib/mi355_compute_ops*.so, source csrc/pybind/register_ops.cc): the pybind11 op
module that exports ``rtp_llm::registerPyModuleOps`` -- the counterpart of the reference's ``librtp_compute_ops`` whose
ops Python reaches as ``rtp_llm.ops.compute_ops.rtp_llm_ops`` (rtp_llm/ops/compute_ops.py:3-4)."""
import importlib.util
import os

_mod = None


def load():
    """-> the ``rtp_llm_ops`` submodule.  Raises if the shim has not been built (python -m rtp_llm_amd.build --pybind)."""
    global _mod
    if _mod is None:
        import torch  # noqa: F401  (libtorch must be loaded first)
        from . import _C
        from .build import pybind_module_path
        _C.lib()                                    # the C-ABI library the shim links against
        path = pybind_module_path()
        if not os.path.exists(path):
            raise _C.Mi355Error(f"{path} not found - build it with `python -m rtp_llm_amd.build --pybind`")
        spec = importlib.util.spec_from_file_location("mi355_compute_ops", path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _mod = m
    return _mod.rtp_llm_ops
