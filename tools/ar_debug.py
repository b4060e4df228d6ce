#!/usr/bin/env python3
"""Debug helper: N processes on one GPU, a sequence of custom all-reduces, prints which rows differ from the rank-order fp32 sum.
usage: ar_debug.py WORLD "T,H;T,H;..." """
import os, socket, sys
import torch, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(rank, world, port, shapes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from rtp_llm_amd import distributed
    torch.cuda.set_device(0); dev = "cuda:0"
    distributed.init_distributed("gloo")
    ar = distributed.CustomAllReduce(max_bytes=300 * 8192 * 2)
    g = torch.Generator().manual_seed(100 + rank)
    msgs = []
    for it, (T, H) in enumerate(shapes):
        x = (torch.randn(T, H, generator=g) * 2).half()
        got = ar.all_reduce(x.to(dev).clone()); torch.cuda.synchronize()
        outs = [torch.empty_like(x) for _ in range(world)]; dist.all_gather(outs, x)
        acc = torch.zeros(T, H)
        for p in outs: acc = acc + p.float()
        bad = (got.cpu() != acc.half())
        rows = bad.any(dim=1).nonzero().flatten().tolist()
        cols = bad.any(dim=0).nonzero().flatten().tolist()
        msgs.append(f"rank {rank} call {it} ({T},{H}): {len(rows)} bad rows {rows[:12]}{'...' if len(rows) > 12 else ''}; bad cols {len(cols)} [{cols[:4]}..{cols[-2:] if cols else ''}] status {ar.status()}")
        dist.barrier()
    q.put("\n".join(msgs)); dist.barrier(); ar.close(); dist.destroy_process_group()

if __name__ == "__main__":
    world = int(sys.argv[1]); shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[2].split(";")]
    with socket.socket() as s: s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, shapes, q)) for r in range(world)]
    [p.start() for p in ps]
    for _ in range(world): print(q.get(timeout=300))
    [p.join(30) for p in ps]
