#!/usr/bin/env python3
"""Per-wave wall_clock64 stamps of one gemm_pc launch (tuning build).  usage: pc_stamps.py [dbg]"""
import ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtp_llm_amd import _C, model, ops
K, N, M, dev = 3584, 37888, 64, "cuda:0"
lib = _C.lib(); lib.mi355_debug_set.argtypes = [C.c_int, C.c_int]
gen = torch.Generator(device=dev).manual_seed(0)
w = model.synth_linear(K, N, "w4", dev, gen).pack(gate_up=True)
x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
st = torch.zeros(256 * 12 * 4, dtype=torch.int64, device=dev)
lib.mi355_debug_ptr.argtypes = [C.c_void_p]; lib.mi355_debug_ptr(st.data_ptr())
lib.mi355_debug_set(7, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for _ in range(3):
    ops.linear(x, w, None, _C.EPI_SILU_MUL)
torch.cuda.synchronize()
s = st.view(256, 12, 4).cpu().double() * 0.01  # us
t0 = s[..., 0].min()
print(f"last wave starts at {s[..., 0].max() - t0:.2f} us; block start spread {(s[:, 0, 0] - t0).max():.2f}")
for name, sl in (("producers", slice(0, 4)), ("consumers", slice(4, 12))):
    q = s[:, sl]
    print(f"{name}: prologue mean {(q[..., 1] - q[..., 0]).mean():.2f} max {(q[..., 1] - q[..., 0]).max():.2f} | loop mean {(q[..., 2] - q[..., 1]).mean():.2f} max {(q[..., 2] - q[..., 1]).max():.2f} | tail mean {(q[..., 3] - q[..., 2]).mean():.2f} max {(q[..., 3] - q[..., 2]).max():.2f}")
print(f"last wave ends at {s[..., 3].max() - t0:.2f} us; per-block duration mean {(s[..., 3].amax(1) - s[..., 0].amin(1)).mean():.2f}")
