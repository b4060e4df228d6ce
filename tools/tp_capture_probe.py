#!/usr/bin/env python3
"""Can the TP decode step (C++ segments + RCCL collectives issued from Python) be captured in one graph on this
torch / RCCL build?  Single process, world-size-1 NCCL group (the all-reduce is an identity, numerics are not the point),
engine built with per-rank shapes of tp=2 and tp_size=2 so that step_tp takes the all-reduce path."""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtp_llm_amd import distributed, model

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", RANK="0", WORLD_SIZE="1")
dist.init_process_group("gloo")
grp = dist.new_group([0], backend="nccl")
distributed.set_tp_group(grp)
dev = "cuda:0"
torch.cuda.set_device(0)
full = model.ModelConfig("probe", 4, 1024, 16, 4, 64, 2048, 4096, max_pos=256)
cfg = full.per_rank(2)
w = model.synth_model(cfg, "w4", dev, seed=1)
w["embedding"] = (torch.randn(full.vocab, cfg.hidden, device=dev) * 0.5).half()
B = 8
eng = model.DecoderEngine(cfg, w, kv_int8=False, page=16, num_blocks=B * 4, max_batch=B, max_seq_len=64, device=dev, tp_size=2,
                          vocab_full=full.vocab)
# step_tp consults distributed.tp_size(): make the world-1 group look like TP so the collectives are really issued
distributed.tp_size = lambda: 2
import rtp_llm_amd.distributed as D
_ar = dist.all_reduce
bt = torch.arange(B * 4, dtype=torch.int32).reshape(B, 4)
eng.set_inputs([1] * B, [0] * B, bt)
for _ in range(2):
    eng.step_tp(B)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        eng.step_tp(B)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"capture OK: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per replayed TP step (4 layers, world-1 RCCL group)")
except Exception as e:  # noqa: BLE001
    print("capture FAILED:", type(e).__name__, str(e).splitlines()[0])
