"""Build libmi355_decode.so (gfx950) in-tree with hipcc.

    python -m rtp_llm_amd.build [--force] [--tuning] [--pybind]

Objects go to rtp_llm_amd/csrc/build/, the library to rtp_llm_amd/lib/.  Only
stale translation units are recompiled (mtime of source + headers).
--tuning additionally builds lib/libmi355_decode_tuning.so (-DMI355_TUNING: process-global experiment
switches for tools/; never loaded by the product path or the tests).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmi355_decode.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

SOURCES = ["gemm.hip", "gemm_smallm.hip", "gemm_wide.hip", "gemm_fullk.hip", "gemm_fullk64.hip", "gemm_splitk64.hip", "gemm_prefill.hip", "attention.hip", "rope_kv.hip", "elementwise.hip", "sampling.hip", "allreduce.hip", "rccl_transport.cpp", "engine.cpp", "error.cpp"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm_fullk.h"), os.path.join(CSRC, "internal.h"), os.path.join(INCLUDE, "mi355_decode.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-result", "-x", "hip"]


def _stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in [src] + HEADERS)


def _compile(name, force, tuning=False):
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, os.path.splitext(name)[0] + ("_tuning.o" if tuning else ".o"))
    if not force and not _stale(src, obj):
        return obj, False
    extra = os.environ.get("MI355_EXTRA_CFLAGS", "").split() if tuning else []   # tuning build only (e.g. -DMI355_FULLK_STAMPS)
    cmd = [HIPCC] + CFLAGS + (["-DMI355_TUNING"] if tuning else []) + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(force=False, verbose=True, tuning=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    LIB = os.path.join(LIBDIR, "libmi355_decode_tuning.so" if tuning else "libmi355_decode.so")
    srcs = SOURCES
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, tuning), srcs))
    objs = [o for o, _ in res]
    rebuilt = any(ch for _, ch in res)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[rtp_llm_amd.build] built {LIB}")
    elif verbose:
        print(f"[rtp_llm_amd.build] up to date: {LIB}")
    return LIB


PYBIND_SRC = os.path.join(CSRC, "pybind", "register_ops.cc")


def pybind_module_path():
    import sysconfig
    return os.path.join(LIBDIR, "mi355_compute_ops" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force=False, verbose=True):
    """The native registration shim (csrc/pybind/register_ops.cc): a pybind11 / torch-extension module exporting
    rtp_llm::registerPyModuleOps on top of the C-ABI library.  Host code only; compiled with hipcc so that the HIP and
    torch-ROCm headers see their usual configuration."""
    import sysconfig
    import torch
    out = pybind_module_path()
    lib = os.path.join(LIBDIR, "libmi355_decode.so")
    deps = [PYBIND_SRC, os.path.join(INCLUDE, "mi355_decode.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        if verbose:
            print(f"[rtp_llm_amd.build] up to date: {out}")
        return out
    if not os.path.exists(lib):
        build(verbose=verbose)
    T = os.path.dirname(torch.__file__)
    cmd = [HIPCC, "-x", "hip", f"--offload-arch={ARCH}", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-DTORCH_EXTENSION_NAME=mi355_compute_ops", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{T}/include", f"-I{T}/include/torch/csrc/api/include", f"-I{sysconfig.get_paths()['include']}",
           PYBIND_SRC, "-o", out, f"-L{T}/lib", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python",
           f"-L{LIBDIR}", "-lmi355_decode", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{T}/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"pybind shim build failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")
    if verbose:
        print(f"[rtp_llm_amd.build] built {out}")
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--tuning" in sys.argv:
        build(force="--force" in sys.argv, tuning=True)
    if "--pybind" in sys.argv:
        build_pybind(force="--force" in sys.argv)
