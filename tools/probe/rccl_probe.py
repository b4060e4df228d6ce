import os, sys, ctypes as C, socket
import torch, torch.multiprocessing as mp
def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    class UID(C.Structure): _fields_ = [("b", C.c_char * 128)]
    uid = UID()
    if rank == 0:
        assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    blobs = [bytes(uid.b) if rank == 0 else None]
    dist.broadcast_object_list(blobs, 0)
    C.memmove(C.byref(uid), blobs[0], 128)
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
    rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
    msg = f"rank {rank}: ncclCommInitRank rc={rc}"
    if rc == 0:
        x = torch.full((8,), float(rank + 1), dtype=torch.float16, device="cuda:0")
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        rc2 = lib.ncclAllReduce(x.data_ptr(), x.data_ptr(), 8, 6, 0, comm, st)
        torch.cuda.synchronize()
        msg += f" allreduce rc={rc2} -> {x[0].item()}"
    q.put(msg)
if __name__ == "__main__":
    world = int(sys.argv[1])
    with socket.socket() as s: s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    for _ in range(world):
        try: print(q.get(timeout=120))
        except Exception as e: print("timeout", e)
    [p.join(10) for p in ps]
    [p.kill() for p in ps if p.is_alive()]
