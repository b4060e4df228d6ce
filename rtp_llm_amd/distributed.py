"""TP collectives with the reference's call shape (rtp_llm/models_py/distributed/collective_torch.py:694-769):
all_reduce(tensor, Group.TP) / all_gather(tensor, Group.TP).  One process per GPU.

Two transports, chosen like the reference does under graph capture (rocm_rccl.py:511-572: its custom one-shot kernel when
the message fits the registered workspace, RCCL otherwise):
  * ``CustomAllReduce`` -- the hand-written one-shot peer-read kernel of csrc/allreduce.hip over IPC-mapped buffers
    (xGMI between GPUs): fp32 rank-order sums, bit-identical on every rank, graph-capturable, optionally fused with the
    residual add + RMSNorm (and the split-K reduce) by the C++ step driver;
  * torch.distributed -- "nccl" is RCCL on ROCm, "gloo" for the CPU tests -- for everything else."""
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


_custom_ar = None


def _C_err(msg):
    from . import _C
    return _C.Mi355Error(msg)


class CustomAllReduce:
    """Host side of csrc/allreduce.hip (the role of TrtllmArFusionHandle + its Python wrapper, base/rocm/trt_allreduce.py:
    51-230): create the context, exchange the IPC handle blobs through the (CPU-capable) process group, open the peers."""

    def __init__(self, max_bytes: int, group=None, rank: Optional[int] = None, world: Optional[int] = None, spin_timeout_ms: Optional[int] = None,
                 verify: bool = True):
        import ctypes as C
        from . import _C
        self._C, self.lib = _C, _C.lib()
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        n = self.lib.mi355_allreduce_handle_bytes()
        blob = C.create_string_buffer(n)
        self.handle = self.lib.mi355_allreduce_create(self.rank, self.world, int(max_bytes), blob)
        err = None if self.handle else "allreduce_create failed: " + self.lib.mi355_last_error().decode()
        blobs = [None] * self.world
        dist.all_gather_object(blobs, None if err else blob.raw, group=group)   # host bytes: works over gloo and over nccl groups
        if err is None and all(b is not None for b in blobs):
            self._all = C.create_string_buffer(b"".join(blobs), n * self.world)
            if self.lib.mi355_allreduce_open(self.handle, self._all) < 0:
                err = "allreduce_open failed: " + self.lib.mi355_last_error().decode()
        # every rank learns whether every rank mapped every peer (this exchange is also the barrier before first use): a
        # failure anywhere raises everywhere, so the callers' fallback (RcclTransport) is taken by all ranks together
        errs = [None] * self.world
        dist.all_gather_object(errs, err, group=group)
        if any(errs):
            self.close()
            raise _C.Mi355Error("; ".join(f"rank {r}: {e}" for r, e in enumerate(errs) if e))
        self.max_bytes = int(max_bytes)
        # ranks that share ONE device (the single-GPU validation runs of the tests) are time-sliced against each other: their kernels
        # wait for a peer whose kernel may not be resident yet, so the 2 s spin bound of a one-process-per-GPU deployment is raised
        keys = [None] * self.world
        dist.all_gather_object(keys, self._device_key(), group=group)
        self.shared_device = len(set(keys)) < len(keys)
        if self.shared_device and self.rank == 0:
            import sys
            print(f"[rtp_llm_amd.distributed] {self.world} ranks on {len(set(keys))} device(s): ranks share a GPU, the all-reduce spin bound is raised to 30 s",
                  file=sys.stderr, flush=True)
        if self.shared_device if spin_timeout_ms is None else True:
            self.set_spin_timeout_ms(30000 if spin_timeout_ms is None else spin_timeout_ms)
        self._group = group
        # The hand-over form must be ONE for the whole group: the flag forms interoperate with each other but not with the granule form, and an
        # environment variable set on a subset of the ranks would leave them spinning on each other.  Every rank proposes what its
        # environment asks for; the group takes the most conservative proposal and every context is set to it explicitly.
        want = ("full-fences" if os.environ.get("MI355_AR_FULL_FENCES") == "1" else
                "ll" if os.environ.get("MI355_AR_LL") == "1" else "write-through")
        wants = [None] * self.world
        dist.all_gather_object(wants, want, group=group)
        agreed = self.PROTOCOLS[max(self.PROTOCOLS.index(w) for w in wants)]
        if len(set(wants)) > 1 and self.rank == 0:
            import sys
            print(f"[rtp_llm_amd.distributed] ranks asked for different all-reduce hand-overs {wants}: the group uses {agreed}", file=sys.stderr, flush=True)
        self.set_protocol(agreed)
        if verify and self.world > 1:
            self._verify_hand_over()

    def _known_answer_round(self, rounds: Optional[int] = None) -> bool:
        """Known-answer collectives on every path that publishes through the hand-over under test -- the one-shot sum (<= 64 rows), the two-shot
        sum (> 64 rows on > 2 ranks: second barrier + result region), the fused residual + RMSNorm form, the all-gather of the hidden dimension and
        the cross-rank arg-max -- `rounds` times, so both buffer parities are used many times, with one rank (rotating) held back by a matmul in
        front of its call so that its peers really wait on its flags.  Integer patterns: every partial sum is exact in fp32 and fp16, so the
        expected bits do not depend on anything but the protocol.  True when this rank saw the expected bits every time."""
        dev = torch.device("cuda", torch.cuda.current_device())
        H = 1024
        rows_max = max(1, self.max_bytes // (H * 2))
        if rounds is None:
            rounds = 6 if self.shared_device else 24               # ranks time-slicing ONE device (tests): every call is a chain of process switches
        if self.shared_device:
            rows_max = min(rows_max, 768 // self.world)            # ... and every block of every rank has to be resident at once
        T1 = min(64, rows_max)
        T2 = min(160, rows_max) if self.world > 2 else 0          # two-shot geometry (one-shot again if the registered buffer is too small for > 64 rows)
        ng = 64                                                     # all-gather slice width
        Tg = max(1, min(16, self.max_bytes // (ng * self.world * 2)))
        ok = True
        slow = torch.randn(1024, 1024, device=dev)
        for it in range(rounds):
            if it % self.world == self.rank:                        # this rank arrives late this round
                for _ in range(4):
                    slow = (slow @ slow).clamp_(-1, 1)
            for T in filter(None, (T1, T2 if it % 3 == 0 else 0)):
                idx = torch.arange(T * H, dtype=torch.int64).reshape(T, H)
                pats = [((idx * (r + 3) + it * 7 + r) % 61 - 30).to(torch.float16) for r in range(self.world)]    # |value| <= 30: sums of 8 ranks exact
                want = sum(p.float() for p in pats).to(torch.float16)
                got = self.all_reduce(pats[self.rank].to(dev).clone())
                ok = ok and torch.equal(got.cpu(), want)
            if it % 4 == 1:      # fused form: residual add + RMSNorm of the sum (weight 1): the residual out must be sum + residual, bit for bit
                idx = torch.arange(T1 * H, dtype=torch.int64).reshape(T1, H)
                pats = [((idx * (r + 5) + it * 3 + r) % 31 - 15).to(torch.float16) for r in range(self.world)]
                res = ((idx * 11 + it) % 17 - 8).to(torch.float16)
                _, res_out = self.all_reduce_add_rmsnorm(pats[self.rank].to(dev), res.to(dev), torch.ones(H, dtype=torch.float16, device=dev), 1e-6)
                ok = ok and torch.equal(res_out.cpu(), (sum(p.float() for p in pats) + res.float()).to(torch.float16))
            if it % 4 == 2:      # all-gather of the hidden dimension
                slices = [((torch.arange(Tg * ng, dtype=torch.int64).reshape(Tg, ng) * (r + 2) + it) % 97 - 48).to(torch.float16) for r in range(self.world)]
                got = self.all_gather_hidden(slices[self.rank].to(dev))
                ok = ok and torch.equal(got.cpu(), torch.cat(slices, dim=1))
            if it % 4 == 3:      # cross-rank arg-max: the winner sits in a rank that rotates with the round
                Bq, Vl = 8, 512
                base = ((torch.arange(Bq * Vl, dtype=torch.int64).reshape(Bq, Vl) * 7 + it) % 101).float()
                win_rank, win_col = (it // 4) % self.world, (it * 37) % Vl
                mine = base.clone()
                if win_rank == self.rank:
                    mine[:, win_col] = 1000.0 + it
                ids = self.argmax(mine.to(dev), self.rank * Vl)
                ok = ok and bool((ids.cpu() == win_rank * Vl + win_col).all())
        torch.cuda.synchronize()
        return ok and self.status() == 0

    PROTOCOLS = ("ll", "write-through", "full-fences")     # mi355_allreduce_set_protocol modes 0 / 1 / 2

    def set_protocol(self, name: str) -> None:
        """"write-through" (default): write-through publishing stores + flags; "ll" (opt-in): data-tagged granules for the <= 64-row calls, the rest as
        write-through; "full-fences": plain stores between system-scope release / acquire fences (rounds 1-4).  Same results."""
        self._C.check(self.lib.mi355_allreduce_set_protocol(self.handle, self.PROTOCOLS.index(name)), "allreduce_set_protocol")
        self.hand_over = name

    def _verify_hand_over(self) -> None:
        """The fence-free hand-overs (granules; write-through stores + flags: csrc/allreduce.hip) were developed with several processes on ONE GPU;
        the first thing a context does on a node is a known-answer check on every rank.  If any rank sees a wrong sum the whole group steps down
        one form (ll -> write-through -> full-fences) and checks again; if the fenced form of rounds 1-4 fails too the constructor raises on
        every rank together."""
        import sys
        for attempt in self.PROTOCOLS[self.PROTOCOLS.index(self.hand_over):]:
            if attempt != self.hand_over:
                self.set_protocol(attempt)
                self._C.check(self.lib.mi355_allreduce_clear_status(self.handle, self._st()), "allreduce_clear_status")   # a timed-out spin of the form before
            oks = [None] * self.world
            dist.all_gather_object(oks, bool(self._known_answer_round()), group=self._group)
            if all(oks):
                return
            if self.rank == 0:
                print(f"[rtp_llm_amd.distributed] all-reduce known-answer check failed with the {attempt} hand-over on ranks "
                      f"{[r for r, o in enumerate(oks) if not o]}", file=sys.stderr, flush=True)
        self.close()
        raise _C_err("CustomAllReduce: the all-reduce known-answer check fails with every hand-over form")

    @staticmethod
    def _device_key():
        import socket
        import os
        idx = torch.cuda.current_device()
        p = torch.cuda.get_device_properties(idx)
        ident = getattr(p, "uuid", None)
        if not ident and getattr(p, "pci_bus_id", None) is not None:
            ident = (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", -1))
        if not ident:
            # a torch build that exposes neither a uuid nor a PCI id: the ordinal inside this process's visible set -- never ONE key for
            # every rank of a host (that would read as "all ranks share a device" and raise the spin bound on a real multi-GPU node)
            ident = ("ordinal", idx, os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")),
                     os.environ.get("ROCR_VISIBLE_DEVICES", ""))
        return (socket.gethostname(), str(ident))

    def set_full_fences(self, on: bool) -> None:
        """Hand-over protocol of later launches: False (default) = write-through publishing stores + drained flags, True = plain stores
        between system-scope release / acquire fences (rounds 1-4).  Same results (mi355_allreduce_set_full_fences)."""
        self._C.check(self.lib.mi355_allreduce_set_full_fences(self.handle, 1 if on else 0), "allreduce_set_full_fences")
        self.hand_over = "full-fences" if on else "write-through"

    def set_spin_timeout_ms(self, ms: int) -> None:
        """Bound of every in-kernel wait for a peer; applies to launches enqueued or captured afterwards."""
        self._C.check(self.lib.mi355_allreduce_set_spin_timeout_ms(self.handle, int(ms)), "allreduce_set_spin_timeout_ms")

    def _st(self):
        return torch.cuda.current_stream().cuda_stream

    def fits(self, t: torch.Tensor) -> bool:
        return (t.is_cuda and t.dtype in (torch.float16, torch.bfloat16) and t.is_contiguous() and t.dim() >= 1 and t.shape[-1] % 8 == 0
                and t.shape[-1] <= 8192 and t.numel() * 2 <= self.max_bytes)

    def all_reduce(self, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = t if out is None else out
        H = t.shape[-1]
        self._C.check(self.lib.mi355_allreduce_sum_dt(self.handle, t.data_ptr(), out.data_ptr(), t.numel() // H, H,
                                                      self._C.ACT_BF16 if t.dtype == torch.bfloat16 else self._C.ACT_F16, self._st()), "allreduce_sum")
        return out

    def all_reduce_add_rmsnorm(self, t, residual, weight, eps):
        """(normed, residual_out) = RMSResNorm(all_reduce(t), residual) in one launch (allreduce_fusion_kernel_1stage)."""
        H = t.shape[-1]
        y, res = torch.empty_like(t), torch.empty_like(t)
        self._C.check(self.lib.mi355_allreduce_fused_dt(self.handle, t.data_ptr(), None, 0, 0, None, residual.data_ptr(), res.data_ptr(),
                                                        weight.data_ptr(), float(eps), t.numel() // H, H, y.data_ptr(),
                                                        self._C.ACT_BF16 if t.dtype == torch.bfloat16 else self._C.ACT_F16, self._st()),
                      "allreduce_fused")
        return y, res

    def all_reduce_published_add_rmsnorm(self, residual, weight, eps):
        """all_reduce_add_rmsnorm for rows the GEMM in front of it published in the registered buffer (ops.linear_publish_img on the same stream, no other call
        of this context in between): (normed, residual_out) (mi355_allreduce_fused_published_dt)."""
        H = residual.shape[-1]
        y, res = torch.empty_like(residual), torch.empty_like(residual)
        self._C.check(self.lib.mi355_allreduce_fused_published_dt(self.handle, residual.data_ptr(), res.data_ptr(), weight.data_ptr(), float(eps),
                                                                  residual.numel() // H, H, y.data_ptr(), 0,
                                                                  self._C.ACT_BF16 if residual.dtype == torch.bfloat16 else self._C.ACT_F16, self._st()),
                      "allreduce_fused_published")
        return y, res

    def set_prefetch(self, t: torch.Tensor = None):
        """The NEXT fused all-reduce launch touches the bytes of `t` (normally the next GEMM's weight shard) with the waves that only
        wait for the peers' flags (mi355_allreduce_set_prefetch); None: off.  No effect on results."""
        self._C.check(self.lib.mi355_allreduce_set_prefetch(self.handle, t.data_ptr() if t is not None else None,
                                                            t.numel() * t.element_size() if t is not None else 0), "allreduce_set_prefetch")

    def all_gather_hidden(self, t: torch.Tensor) -> torch.Tensor:
        """[T, n] column slice per rank -> [T, n * world], slices side by side in rank order (the all_gather + transpose of the
        reference's hidden-split embedding, modules/base/common/embedding.py:50-58)."""
        T, n = t.shape
        out = torch.empty(T, n * self.world, dtype=t.dtype, device=t.device)
        self._C.check(self.lib.mi355_allgather_hidden(self.handle, t.data_ptr(), out.data_ptr(), T, n, self._st()), "allgather_hidden")
        return out

    def argmax(self, logits_local: torch.Tensor, vocab_offset: int) -> torch.Tensor:
        B, V = logits_local.shape
        ids = torch.empty(B, dtype=torch.int32, device=logits_local.device)
        ws = torch.empty(B * 64 * 8, dtype=torch.uint8, device=logits_local.device)
        self._C.check(self.lib.mi355_allreduce_argmax(self.handle, logits_local.data_ptr(), B, V, V, int(vocab_offset), ids.data_ptr(),
                                                      None, ws.data_ptr(), ws.numel(), self._st()), "allreduce_argmax")
        return ids

    def status(self) -> int:
        """0 = healthy; != 0: a peer did not arrive within the kernel's spin bound (results invalid).  Synchronises."""
        return int(self.lib.mi355_allreduce_status(self.handle, self._st()))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mi355_allreduce_destroy(self.handle)
            self.handle = None


class RcclTransport:
    """RCCL communicator behind ``mi355_collective_t`` (csrc/rccl_transport.cpp): the transport the C++ step falls back to
    when CustomAllReduce cannot map its peers -- ncclAllReduce / ncclAllGather enqueued on the step's stream, so they are
    captured into the step graph like the kernels around them (the reference: rocm_rccl.py:511-572).  The unique id made
    by rank 0 travels through the (CPU-capable) process group, as the reference's does through its TCPStore
    (rocm_rccl.py:150-260).  librccl is the copy torch ships, shared with torch.distributed's "nccl" backend."""

    def __init__(self, group=None, rank: Optional[int] = None, world: Optional[int] = None, lib_path: Optional[str] = None):
        import ctypes as C
        from . import _C
        self._C, self.lib = _C, _C.lib()
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        if lib_path is None:
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            lib_path = cand if os.path.exists(cand) else ""
        path = lib_path.encode()
        n = self.lib.mi355_rccl_unique_id_bytes()
        blob = C.create_string_buffer(n)
        # every rank learns about a failure of any rank before anyone raises (as CustomAllReduce does): rank 0 broadcasts the id OR its
        # error, the open status is all-gathered -- a rank that raised alone would leave its peers blocked in the broadcast, or alone on
        # another transport
        err = None
        if self.rank == 0:
            rc = self.lib.mi355_rccl_unique_id(path, blob)
            if rc < 0:
                err = "rccl_unique_id failed: " + self.lib.mi355_last_error().decode(errors="replace")
        if self.world > 1:
            box = [(blob.raw, err)]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            raw, err = box[0]
            blob = C.create_string_buffer(raw, n)
        if err is not None:
            raise _C.Mi355Error(err)
        self.handle = self.lib.mi355_rccl_open(path, blob, self.rank, self.world)
        mine = None if self.handle else "rank %d: rccl_open failed: %s" % (self.rank, self.lib.mi355_last_error().decode(errors="replace"))
        errs = [mine]
        if self.world > 1:
            errs = [None] * self.world
            dist.all_gather_object(errs, mine, group=group)
        bad = [e for e in errs if e]
        if bad:
            self.close()
            raise _C.Mi355Error("; ".join(bad))
        self.collective = _C.Collective()
        _C.check(self.lib.mi355_rccl_collective(self.handle, C.byref(self.collective)), "rccl_collective")

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM of an fp16 / bf16 tensor over the ranks, on the current stream."""
        if not (t.is_cuda and t.dtype in (torch.float16, torch.bfloat16) and t.is_contiguous()):
            raise ValueError("RcclTransport.all_reduce: contiguous fp16 / bf16 device tensor expected")
        fn = self.collective.all_reduce_bf16 if t.dtype == torch.bfloat16 else self.collective.all_reduce_f16
        rc = fn(self.collective.ctx, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise self._C.Mi355Error("ncclAllReduce failed: " + self.lib.mi355_last_error().decode())
        return t

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mi355_rccl_close(self.handle)
            self.handle = None


def set_custom_all_reduce(ar: Optional[CustomAllReduce]) -> None:
    """Route fitting TP all-reduces through the hand-written kernel (like the reference's capture path does)."""
    global _custom_ar
    _custom_ar = ar


def all_reduce(tensor: torch.Tensor, group: Group = Group.TP, inplace: bool = True) -> torch.Tensor:
    """SUM over the TP ranks (C1/C2 of SURVEY 2.3: after O-proj and down-proj)."""
    if tp_size() == 1:
        return tensor
    t = tensor if inplace else tensor.clone()
    if _custom_ar is not None and _custom_ar.fits(t):
        return _custom_ar.all_reduce(t)
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
