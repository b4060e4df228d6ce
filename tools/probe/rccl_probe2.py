"""Does RCCL bootstrap on this box at all?  (1) torch.distributed "nccl" world 1, (2) rtp_llm_amd.distributed.RcclTransport world 1."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import torch.distributed as dist
which = sys.argv[1]
torch.cuda.set_device(0)
if which == "torch":
    try:
        dist.init_process_group("nccl")
        x = torch.ones(8, device="cuda:0", dtype=torch.float16)
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print("torch nccl world-1 all_reduce ok", x[0].item(), flush=True)
    except Exception:
        traceback.print_exc()
else:
    try:
        from rtp_llm_amd import distributed
        t = distributed.RcclTransport(rank=0, world=1)
        x = torch.full((16,), 3.0, device="cuda:0", dtype=torch.float16)
        t.all_reduce(x)
        torch.cuda.synchronize()
        print("RcclTransport world-1 all_reduce ok", x[0].item(), flush=True)
    except Exception:
        traceback.print_exc()
