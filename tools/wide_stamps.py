#!/usr/bin/env python3
"""Where the time of one gemm_wide launch goes: per-wave wall_clock64 stamps (100 MHz) at kernel start, after the
prologue, after the main loop and at the end (debug variant 4).  usage: wide_stamps.py [K]"""
import ctypes as C, os, sys
import torch
os.environ["MI355_TUNING_LIB"] = "1"   # experiment switches live in the tuning build only (python -m rtp_llm_amd.build --tuning)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import _C, model, ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
N, M, dev = 37888, 64, "cuda:0"
lib = _C.lib()
gen = torch.Generator(device=dev).manual_seed(0)
w = model.synth_linear(K, N, "w4", dev, gen).pack(gate_up=True)
x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
st = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
lib.mi355_debug_ptr.argtypes = [C.c_void_p]
lib.mi355_debug_ptr(st.data_ptr())
lib.mi355_debug_set(0, int(sys.argv[2]) if len(sys.argv) > 2 else 4)   # 7: the same stamps without the loop's memory traffic
for _ in range(3):
    ops.linear(x, w, None, _C.EPI_SILU_MUL)
torch.cuda.synchronize()
s = st.view(256, 8, 4).cpu().double() * 0.01  # us
t0 = s[..., 0].min()
print(f"K={K}: first wave starts at 0, last wave starts at {s[..., 0].max() - t0:.2f} us")
print(f"prologue  (start -> first unit): mean {(s[..., 1] - s[..., 0]).mean():.2f} us, max {(s[..., 1] - s[..., 0]).max():.2f}")
print(f"main loop                      : mean {(s[..., 2] - s[..., 1]).mean():.2f} us, max {(s[..., 2] - s[..., 1]).max():.2f}")
print(f"merge + epilogue               : mean {(s[..., 3] - s[..., 2]).mean():.2f} us, max {(s[..., 3] - s[..., 2]).max():.2f}")
print(f"last wave ends at {s[..., 3].max() - t0:.2f} us")
