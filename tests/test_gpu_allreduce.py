"""The hand-written one-shot all-reduce (csrc/allreduce.hip) with TWO PROCESSES ON ONE GPU: IPC handles map a peer's
buffers across processes on the same device exactly as across devices, so everything except the xGMI hop itself is
exercised -- handle exchange, per-block flag barrier, double buffering over many calls, graph capture + replay, the fused
residual + RMSNorm epilogue, the cross-rank greedy argmax, and the C++ tensor-parallel step built on them.
Checks: bit-exact against the fp32 rank-order sum (the reference's numerics, trtllm_allreduce_fusion.cu:228-246), results
identical on both ranks, fused == unfused bit for bit, TP engine logits vs the unsplit oracle."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gather_cpu(t, world):
    import torch.distributed as dist
    outs = [torch.empty_like(t.cpu()) for _ in range(world)]
    dist.all_gather(outs, t.cpu())
    return outs


def _worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if mode.endswith("_ff"):      # the hand-over of rounds 1-4 (plain stores between system-scope release / acquire fences) instead of the
        os.environ["MI355_AR_FULL_FENCES"] = "1"   # fence-free forms of round 5: every form passes the same checks
        mode = mode[:-3]
    elif mode.endswith("_ll"):    # data-tagged granules (LL) for the <= 64-row calls instead of publishing stores + flags: opt-in form
        os.environ["MI355_AR_LL"] = "1"
        mode = mode[:-3]
    try:
        import torch.distributed as dist
        from oracle import oracle
        from rtp_llm_amd import _C, distributed, model, ops
        dev = "cuda:0"
        torch.cuda.set_device(0)
        torch.set_num_threads(max(1, min(32, (os.cpu_count() or 8) // world)))    # `world` processes on one host: no oversubscription of the CPU side (weight synthesis, oracle)
        distributed.init_distributed("gloo")
        # the ranks share one GPU: CustomAllReduce notices and raises the spin bound to 30 s (time-slicing); the timeout case keeps 2 s
        ar = distributed.CustomAllReduce(max_bytes=300 * 8192 * 2, spin_timeout_ms=2000 if mode == "timeout" else None)
        assert ar.shared_device
        g = torch.Generator().manual_seed(100 + rank)
        # All ranks share ONE GPU here: block b of a rank spins until block b of every peer has arrived, so every block of every
        # rank must be resident at once (on a real node each rank has its own 256 CUs).  Keep world x grid within what one GPU
        # holds (~1000 blocks of 512 threads; 8 ranks x 130 rows timed out in the bounded spin): cap the row counts.
        cap = lambda T: min(T, 768 // world)
        if mode == "kernels":
            for it, (T, H) in enumerate([(1, 3584), (5, 3584), (64, 3584), (64, 8192), (cap(300), 8192), (7, 896), (64, 3584), (64, 3584)]):
                x = (torch.randn(T, H, generator=g) * 2).half()
                got = ar.all_reduce(x.to(dev).clone())
                torch.cuda.synchronize()
                parts = _gather_cpu(x, world)
                acc = torch.zeros(T, H)
                for p in parts:                       # fp32, rank order, one rounding
                    acc = acc + p.float()
                ref = acc.half()
                assert torch.equal(got.cpu(), ref), (it, T, H, float((got.cpu().float() - ref.float()).abs().max()))
                both = _gather_cpu(got.cpu(), world)
                assert all(torch.equal(both[0], o) for o in both[1:])
            # fused: all-reduce + residual + RMSNorm == all_reduce_sum followed by add_rmsnorm, bit for bit
            T, H = 33, 3584
            x, res = (torch.randn(T, H, generator=g)).half().to(dev), torch.randn(T, H, generator=torch.Generator().manual_seed(7)).half().to(dev)
            w = (1 + 0.1 * torch.randn(H, generator=torch.Generator().manual_seed(8))).half().to(dev)
            y, r_out = ar.all_reduce_add_rmsnorm(x, res, w, 1e-6)
            s = ar.all_reduce(x.clone())
            y2, r2 = ops.add_rmsnorm(s, res, w, 1e-6)
            assert torch.equal(y, y2) and torch.equal(r_out, r2)
            # in-launch prefetch: the waiting waves touch a 6 MB range; one launch only; numbers unchanged
            junk = torch.randint(0, 2 ** 31 - 1, (6 << 18,), dtype=torch.int32, device=dev)
            ar.set_prefetch(junk)
            y3, r3 = ar.all_reduce_add_rmsnorm(x, res, w, 1e-6)
            y4, r4 = ar.all_reduce_add_rmsnorm(x, res, w, 1e-6)
            assert torch.equal(y3, y2) and torch.equal(r3, r2) and torch.equal(y4, y2) and torch.equal(r4, r2)
            # graph capture + replay: epochs advance on the device
            xs = (torch.randn(16, 3584, generator=g)).half().to(dev)
            out = torch.empty_like(xs)
            torch.cuda.synchronize(); dist.barrier()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ar.all_reduce(xs, out); ar.all_reduce(xs, out)      # warm-up, same parity sequence on both ranks
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    ar.all_reduce(xs, out)
                    ar.all_reduce(out, out)                          # chained: sum of sums
                for rep in range(5):
                    gr.replay()
                torch.cuda.synchronize()
            parts = _gather_cpu(xs.cpu(), world)
            acc = torch.zeros_like(parts[0], dtype=torch.float32)
            for p_ in parts:
                acc = acc + p_.float()
            s1 = acc.half()
            acc = torch.zeros_like(acc)
            for _ in range(world):
                acc = acc + s1.float()
            assert torch.equal(out.cpu(), acc.half())
            # all-gather of the hidden dimension (hidden-split embedding): slices side by side in rank order, interleaved with
            # all-reduces of other shapes on the same buffers, eager and replayed
            for it, (T, n) in enumerate([(1, 1792), (5, 448), (64, 1792), (cap(300), 8192 // world), (7, 8), (64, 1792)]):
                xs_ = (torch.randn(T, n, generator=g)).half()
                got = ar.all_gather_hidden(xs_.to(dev))
                if it % 2:
                    ar.all_reduce(torch.ones(3, 3584, dtype=torch.float16, device=dev))
                torch.cuda.synchronize()
                ref = torch.cat(_gather_cpu(xs_, world), dim=1)
                assert torch.equal(got.cpu(), ref), (it, T, n)
            ng = 3584 // world
            xg = (torch.randn(16, ng, generator=g)).half().to(dev)
            og, osum = torch.empty(16, 3584, dtype=torch.float16, device=dev), torch.empty(16, 3584, dtype=torch.float16, device=dev)
            torch.cuda.synchronize(); dist.barrier()
            st2 = torch.cuda.Stream()
            with torch.cuda.stream(st2):
                lib, hd = ar.lib, ar.handle
                run = lambda: (_C.check(lib.mi355_allgather_hidden(hd, xg.data_ptr(), og.data_ptr(), 16, ng, st2.cuda_stream), "allgather"),
                               _C.check(lib.mi355_allreduce_sum(hd, og.data_ptr(), osum.data_ptr(), 16, 3584, st2.cuda_stream), "allreduce"))
                run(); torch.cuda.synchronize()
                gr2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr2, stream=st2):
                    run()
                for rep in range(4):
                    gr2.replay()
                torch.cuda.synchronize()
            refg = torch.cat(_gather_cpu(xg.cpu(), world), dim=1)
            acc = torch.zeros_like(refg, dtype=torch.float32)
            for _ in range(world):
                acc = acc + refg.float()
            assert torch.equal(og.cpu(), refg) and torch.equal(osum.cpu(), acc.half())
            with pytest.raises(_C.Mi355Error):
                ar.all_gather_hidden(torch.zeros(4, 12, dtype=torch.float16, device=dev))       # not whole 16-byte vectors
            # cross-rank greedy argmax incl. a tie across the rank boundary (lowest global index wins)
            V = 5000
            lg = torch.randn(9, V, generator=g)
            lg[3, 17] = 50.0                                          # same max on every rank -> rank 0's column 17
            lg[4, 100 + rank] = 60.0 + rank                           # the last rank holds the larger value
            ids = ar.argmax(lg.to(dev), rank * V)
            full = torch.cat(_gather_cpu(lg, world), dim=1)
            assert torch.equal(ids.cpu(), torch.argmax(full, -1).int()), (ids.cpu(), torch.argmax(full, -1))
            assert ar.status() == 0
        elif mode == "publish":
            # round 6: a row-parallel shard written by its full-K GEMM straight into the registered buffer (ops.linear_publish_img, gemm_fullk64 FK_PUB) +
            # the fused all-reduce that pulls it (mi355_allreduce_fused_published_dt): against the oracle's linear per rank, summed in fp32 in rank order,
            # residual add and RMSNorm; O / down shapes of Qwen2-7B at tp = world and Llama-3-70B's at tp 8 widths, 1-64 rows; eager, interleaved with
            # other calls of the context (parities advance), and replayed from a graph (the GEMM reads the parity off the device-resident counters)
            from rtp_llm_amd import quant
            shapes = [(3584 // world if world <= 4 else 512, 3584), (1024, 8192), (4736 if world == 4 else 3584, 3584)]
            for si, (K, N) in enumerate(shapes):
                gw = torch.Generator().manual_seed(500 + 10 * si + rank)
                lin = model.synth_linear(K, N, "w4", "cpu", gw, zeros="centered")
                packed = lin.pack()
                W = oracle.dequant_groupwise(lin.q, lin.z_eff, lin.scales, lin.group_size)
                wd = packed.to(dev) if hasattr(packed, "to") else packed
                gamma = (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(9))).half()
                for T in (64, 33, 16, 5, 1):
                    x = (torch.randn(T, K, generator=gw) * 0.5).half()
                    res = torch.randn(T, N, generator=torch.Generator().manual_seed(70 + T)).half()
                    ok = ops.linear_publish_img(ops.act_image_pack(x.to(dev)), wd, ar)
                    assert ok, (K, N, T)
                    y, r_out = ar.all_reduce_published_add_rmsnorm(res.to(dev), gamma.to(dev), 1e-6)
                    torch.cuda.synchronize()
                    mine = oracle.linear(x, W)                                     # fp16 [T, N]: this rank's shard of the sum
                    parts = _gather_cpu(mine, world)
                    acc = torch.zeros(T, N)
                    for p_ in parts:
                        acc = acc + p_.float()
                    ref_res = (acc.half().float() + res.float()).half()
                    ref_y = oracle.rmsnorm(ref_res, gamma, 1e-6)
                    # the GEMM's fp32 sums differ from the oracle's in order: 1e-2 like every linear; the ranks must agree bit for bit
                    assert torch.allclose(r_out.cpu().float(), ref_res.float(), atol=1e-2 * world, rtol=1e-2), (K, N, T, float((r_out.cpu().float() - ref_res.float()).abs().max()))
                    assert torch.allclose(y.cpu().float(), ref_y.float(), atol=5e-2, rtol=5e-2), (K, N, T)
                    both = _gather_cpu(r_out.cpu(), world)
                    assert all(torch.equal(both[0], o) for o in both[1:])
                    if T % 2:                                                       # an unrelated call in between: the counters of its blocks advance
                        ar.all_reduce(torch.ones(3, 3584, dtype=torch.float16, device=dev))
                # published == slab path bit for bit?  Not required (different summation order inside the GEMM); what is required: same as the x_f16 form
                # fed with the GEMM's own rows.  Check through a graph: 6 replays of (publish GEMM -> published all-reduce), outputs constant
                T = 24
                x = (torch.randn(T, K, generator=gw) * 0.5).half().to(dev)
                img = ops.act_image_pack(x)
                res = torch.randn(T, N, generator=torch.Generator().manual_seed(71)).half().to(dev)
                gam = gamma.to(dev)
                y0, r0 = torch.empty_like(res), torch.empty_like(res)
                torch.cuda.synchronize(); dist.barrier()
                stp = torch.cuda.Stream()
                with torch.cuda.stream(stp):
                    def run_once():
                        assert ops.linear_publish_img(img, wd, ar)
                        _C.check(ar.lib.mi355_allreduce_fused_published_dt(ar.handle, res.data_ptr(), r0.data_ptr(), gam.data_ptr(), 1e-6, T, N, y0.data_ptr(), 0,
                                                                           _C.ACT_F16, stp.cuda_stream), "published")
                    run_once(); torch.cuda.synchronize()
                    want_r, want_y = r0.clone(), y0.clone()
                    grp = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(grp, stream=stp):
                        run_once()
                    for rep in range(7):                                            # odd count: both parities, and the counters end where an eager call expects them
                        r0.zero_(); y0.zero_()
                        grp.replay()
                        torch.cuda.synchronize()
                        assert torch.equal(r0, want_r) and torch.equal(y0, want_y), (K, N, rep)
            assert ar.status() == 0
        elif mode == "engine":
            cfg = model.ModelConfig("tiny-tp", 2, 512, 8, 2, 64, 768, 1024, max_pos=256)
            w = model.synth_model(cfg, "w4", "cpu", seed=21, zeros="centered")
            V = cfg.vocab
            layers = [model.split_layer_tp(L, cfg, world, rank) for L in w["layers"]]
            head = w["lm_head"].cols(rank * (V // world), (rank + 1) * (V // world))
            shard = {"layers": layers, "embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": head}
            B, page = 5, 16
            eng = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page, num_blocks=B * 2,
                                      max_batch=B, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
            eng.attach_allreduce(ar, rank * (V // world))
            dense = lambda c: (c.w.float() if c.kind == "fp16" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size))
            ow = {"embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": dense(w["lm_head"]),
                  "layers": [{"input_norm": L["input_norm"], "post_norm": L["post_norm"], "qkv_bias": L["qkv_bias"],
                              **{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}
            odec = oracle.OracleDecoder({**cfg.__dict__}, ow)
            okv = oracle.OracleKV(cfg.num_layers, B, False)
            bt = torch.arange(B * 2, dtype=torch.int32).reshape(B, 2)
            tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
            eng.set_inputs(tok.tolist(), [0] * B, bt)
            dist.barrier()
            eng.capture(B)                                            # the WHOLE tp step (collectives included) as one hipGraph
            for step in range(5):
                pos = torch.full((B,), step, dtype=torch.int32)
                _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
                eng.replay(B, 1)
                torch.cuda.synchronize()
                full = torch.cat(_gather_cpu(eng.logits[:B].cpu(), world), dim=1)
                assert torch.allclose(full, ref, atol=1e-2, rtol=1e-2), (step, float((full - ref).abs().max()))
                assert torch.equal(eng.positions[:B].cpu(), pos + 1)
                mine = eng.token_ids[:B].cpu()
                both = _gather_cpu(mine, world)
                assert torch.equal(both[0], both[1])
                assert torch.equal(mine, torch.argmax(full, -1).int())   # cross-rank argmax == argmax of the gathered row
                tok = oracle.greedy(ref)
                eng.token_ids[:B].copy_(tok)
            assert ar.status() == 0 and eng.oob_count() == 0
            # hidden-split embedding (the reference's TP layout of the table): lookup of this rank's columns + all-gather inside
            # the captured step == the replicated table, bit for bit
            shard2 = {**shard, "embedding": model.split_embedding_tp(w["embedding"], world, rank)}
            eng2 = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard2, dev), kv_int8=False, page=page, num_blocks=B * 2,
                                       max_batch=B, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
            eng2.attach_allreduce(ar, rank * (V // world))
            eng2.set_embedding_split(True)
            eng3 = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page, num_blocks=B * 2,
                                       max_batch=B, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
            eng3.attach_allreduce(ar, rank * (V // world))
            # in-launch prefetch (MI355_PF_TP_INLAUNCH: the waiting waves of each fused all-reduce launch request the next shard):
            # off by default, no effect on the numbers
            eng4 = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page, num_blocks=B * 2,
                                       max_batch=B, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
            eng4.attach_allreduce(ar, rank * (V // world))
            eng4.set_weight_prefetch(_C.PF_TP_INLAUNCH)
            tok0 = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(4), dtype=torch.int32)
            for e in (eng2, eng3, eng4):
                e.set_inputs(tok0.tolist(), [0] * B, bt)
            dist.barrier()
            eng2.capture(B); eng3.capture(B); eng4.capture(B)
            for step in range(4):
                eng2.replay(B, 1); eng3.replay(B, 1); eng4.replay(B, 1)
                torch.cuda.synchronize()
                assert torch.equal(eng2.logits[:B], eng3.logits[:B]) and torch.equal(eng2.token_ids[:B], eng3.token_ids[:B]), step
                assert torch.equal(eng4.logits[:B], eng3.logits[:B]) and torch.equal(eng4.token_ids[:B], eng3.token_ids[:B]), step
            assert ar.status() == 0
        elif mode == "transport":
            # the external-transport form of the TP step (mi355_decoder_attach_collective: the RCCL fallback).  RCCL cannot
            # bootstrap in this sandbox, so the transport under test is a stub with the same contract whose callbacks enqueue
            # the IPC kernels: local fold -> in-place all-reduce -> residual + norm, pair all-gather for greedy; captured.
            from rtp_llm_amd import _C

            class StubTransport:
                def __init__(self, ar):
                    lib = _C.lib()
                    self.H = 512

                    def all_reduce(ctx, buf, count, stream):
                        return lib.mi355_allreduce_sum(ar.handle, buf, buf, count // self.H, self.H, stream)

                    def all_gather(ctx, send, recv, nbytes, stream):
                        assert nbytes % 16 == 0
                        return lib.mi355_allgather_hidden(ar.handle, send, recv, 1, nbytes // 2, stream)
                    self._fns = (_C.ALL_REDUCE_FN(all_reduce), _C.ALL_GATHER_FN(all_gather))
                    self.collective = _C.Collective(None, self._fns[0], _C.ALL_REDUCE_FN(), self._fns[1], ar.rank, ar.world)

            cfg = model.ModelConfig("tiny-tp", 2, 512, 8, 2, 64, 768, 1024, max_pos=256)
            w = model.synth_model(cfg, "w4", "cpu", seed=22, zeros="centered")
            V = cfg.vocab
            layers = [model.split_layer_tp(L, cfg, world, rank) for L in w["layers"]]
            head = w["lm_head"].cols(rank * (V // world), (rank + 1) * (V // world))
            shard = {"layers": layers, "embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": head}
            B, page = 6, 16
            mk = lambda mb: model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page,
                                                num_blocks=mb * 2, max_batch=mb, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
            eng, ref_eng = mk(B), mk(B)
            with pytest.raises(_C.Mi355Error):
                eng.step(B)                                           # tp > 1 with nothing attached
            eng.attach_collective(StubTransport(ar), rank * (V // world))
            ref_eng.attach_allreduce(ar, rank * (V // world))         # the fused IPC step: same sums, same rounding points
            with pytest.raises(_C.Mi355Error):
                eng.step_tp(B)
            bt = torch.arange(B * 2, dtype=torch.int32).reshape(B, 2)
            tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
            for e in (eng, ref_eng):
                e.set_inputs(tok.tolist(), [0] * B, bt)
            dist.barrier()
            eng.capture(B); ref_eng.capture(B)
            for step in range(5):
                eng.replay(B, 1); ref_eng.replay(B, 1)
                torch.cuda.synchronize()
                a, b = eng.logits[:B].cpu(), ref_eng.logits[:B].cpu()
                assert torch.allclose(a, b, atol=2e-3, rtol=2e-3), (step, float((a - b).abs().max()))
                full = torch.cat(_gather_cpu(a, world), dim=1)
                mine = eng.token_ids[:B].cpu()
                assert torch.equal(mine, torch.argmax(full, -1).int()), step
                assert torch.equal(eng.positions[:B].cpu(), torch.full((B,), step + 1, dtype=torch.int32))
                ref_eng.token_ids[:B].copy_(eng.token_ids[:B])
            # eager steps take the same path
            eng.step(B)
            torch.cuda.synchronize()
            assert torch.equal(eng.positions[:B].cpu(), torch.full((B,), 6, dtype=torch.int32))
            assert ar.status() == 0 and eng.oob_count() == 0
        elif mode == "bf16":
            # the collectives and the TP step with bf16 activations: bf16 copies cross the ranks, sums stay fp32 in rank order
            BF = torch.bfloat16
            for T, H in ((5, 512), (64, 3584), (cap(200), 1024)):
                x = (torch.randn(T, H, generator=g) * 0.5).to(BF)
                y = ar.all_reduce(x.to(dev).clone())
                torch.cuda.synchronize()
                parts = _gather_cpu(x, world)
                ref = sum((p.float() for p in parts[1:]), parts[0].float()).to(BF)     # rank order, fp32, one rounding
                assert torch.equal(y.cpu(), ref), (T, H)
            cfg = model.ModelConfig("tiny-tp", 2, 512, 8, 2, 64, 768, 1024, max_pos=256)
            w = model.synth_model(cfg, "w4", "cpu", seed=24, zeros="centered")
            V = cfg.vocab
            layers = [model.split_layer_tp(L, cfg, world, rank) for L in w["layers"]]
            head = w["lm_head"].cols(rank * (V // world), (rank + 1) * (V // world))
            shard = {"layers": layers, "embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": head}
            B, page = 5, 16
            eng = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page, num_blocks=B * 2,
                                      max_batch=B, max_seq_len=32, device=dev, tp_size=world, vocab_full=V, dtype=BF)
            eng.attach_allreduce(ar, rank * (V // world))
            bf = lambda t: None if t is None else t.to(BF)
            dense = lambda c: (c.w.to(BF).float() if c.kind == "fp16" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size))
            ow = {"embedding": bf(w["embedding"]), "final_norm": bf(w["final_norm"]), "lm_head": dense(w["lm_head"]),
                  "layers": [{"input_norm": bf(L["input_norm"]), "post_norm": bf(L["post_norm"]), "qkv_bias": bf(L["qkv_bias"]),
                              **{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}
            odec = oracle.OracleDecoder({**cfg.__dict__}, ow)
            okv = oracle.OracleKV(cfg.num_layers, B, False)
            bt = torch.arange(B * 2, dtype=torch.int32).reshape(B, 2)
            tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
            eng.set_inputs(tok.tolist(), [0] * B, bt)
            dist.barrier()
            eng.capture(B)
            for step in range(5):
                pos = torch.full((B,), step, dtype=torch.int32)
                _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
                eng.replay(B, 1)
                torch.cuda.synchronize()
                full = torch.cat(_gather_cpu(eng.logits[:B].cpu(), world), dim=1)
                assert torch.allclose(full, ref, atol=3e-2, rtol=3e-2), (step, float((full - ref).abs().max()))
                mine = eng.token_ids[:B].cpu()
                assert torch.equal(mine, torch.argmax(full, -1).int())
                tok = oracle.greedy(ref)
                eng.token_ids[:B].copy_(tok)
            assert ar.status() == 0 and eng.oob_count() == 0
        elif mode == "twoshot":
            # world = 3 on one GPU: tensors of more than 64 rows take the two-shot form (rank r reduces rows r, r + 3, ...; second
            # flag barrier; every row fetched once from its owner) -- same numbers as the one-shot: fp32 sum in rank order, one rounding
            for it, (T, H) in enumerate([(65, 3584), (cap(100), 8192), (5, 3584), (cap(130), 896), (cap(100), 8192), (64, 3584), (67, 3584)]):
                x = (torch.randn(T, H, generator=g) * 2).half()
                got = ar.all_reduce(x.to(dev).clone())
                torch.cuda.synchronize()
                parts = _gather_cpu(x, world)
                acc = torch.zeros(T, H)
                for p in parts:
                    acc = acc + p.float()
                assert torch.equal(got.cpu(), acc.half()), (it, T, H)
                both = _gather_cpu(got.cpu(), world)
                assert all(torch.equal(both[0], o) for o in both[1:])
            T, H = cap(100), 3584                                     # fused epilogue on the two-shot path == unfused composition
            x, res = (torch.randn(T, H, generator=g)).half().to(dev), torch.randn(T, H, generator=torch.Generator().manual_seed(7)).half().to(dev)
            w = (1 + 0.1 * torch.randn(H, generator=torch.Generator().manual_seed(8))).half().to(dev)
            y, r_out = ar.all_reduce_add_rmsnorm(x, res, w, 1e-6)
            s = ar.all_reduce(x.clone())
            y2, r2 = ops.add_rmsnorm(s, res, w, 1e-6)
            assert torch.equal(y, y2) and torch.equal(r_out, r2)
            xs_ = (torch.randn(70, 1200 if world == 3 else 1024, generator=g)).half()   # and the all-gather with `world` slices
            got = ar.all_gather_hidden(xs_.to(dev))
            torch.cuda.synchronize()
            assert torch.equal(got.cpu(), torch.cat(_gather_cpu(xs_, world), dim=1))
            # replayed: two-shot then one-shot on the same buffers
            xa, oa, ob = (torch.randn(80, 3584, generator=g)).half().to(dev), torch.empty(80, 3584, dtype=torch.float16, device=dev), torch.empty(8, 3584, dtype=torch.float16, device=dev)   # 80 rows: two-shot at every world > 2
            torch.cuda.synchronize(); dist.barrier()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ar.all_reduce(xa, oa); ar.all_reduce(oa[:8].contiguous(), ob)
                torch.cuda.synchronize()
                xb = oa[:8].contiguous()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    ar.all_reduce(xa, oa)
                    ar.all_reduce(xb, ob)
                for rep in range(4):
                    gr.replay()
                torch.cuda.synchronize()
            parts = _gather_cpu(xa.cpu(), world)
            acc = torch.zeros_like(parts[0], dtype=torch.float32)
            for p_ in parts:
                acc = acc + p_.float()
            s1 = acc.half()
            assert torch.equal(oa.cpu(), s1)
            acc8 = torch.zeros(8, 3584)
            for _ in range(world):
                acc8 = acc8 + s1[:8].float()
            assert torch.equal(ob.cpu(), acc8.half())
            assert ar.status() == 0
        elif mode == "mixed":
            # calls of DIFFERENT widths back to back on one context, no host synchronisation in between --
            # every byte of the registered buffers keeps one owner block whatever the geometry (slots, allreduce.hip), so a
            # fast rank's next call can never overwrite rows a slow peer is still reading
            shapes = [(64, 8192), (7, 896), (33, 3584), (1, 8192), (64, 3584), (5, 896)]
            xs = [(torch.randn(T, H, generator=g) * 2).half() for T, H in shapes]
            dxs = [x.to(dev) for x in xs]
            outs = [torch.empty_like(x) for x in dxs]
            torch.cuda.synchronize(); dist.barrier()
            for rep in range(40):
                if rank == 1 and rep % 7 == 3:
                    torch.cuda._sleep(2_000_000)                      # one rank lags: its peers run up to one call ahead
                for x, o in zip(dxs, outs):
                    ar.all_reduce(x, o)
            torch.cuda.synchronize()
            for x, o in zip(xs, outs):
                acc = torch.zeros_like(x, dtype=torch.float32)
                for p_ in _gather_cpu(x, world):
                    acc = acc + p_.float()
                assert torch.equal(o.cpu(), acc.half()), tuple(x.shape)
            assert ar.status() == 0
        elif mode == "engine70":
            # tp = world with Llama-3-70B's per-rank attention shape (64 q / 8 kv heads of 128 -> 8 q / 1 kv per rank at tp 8;
            # BASELINE configs[3]): 2 layers at hidden 8192 (FFN and vocabulary cut down to keep the CPU oracle quick), the WHOLE
            # step incl. both fused all-reduces per layer and the cross-rank argmax captured as one hipGraph per rank, against
            # the unsplit oracle
            cfg = model.ModelConfig("llama70b-heads", 2, 8192, 64, 8, 128, 4096, 2048, max_pos=256, qkv_bias=False)
            w = model.synth_model(cfg, "w4", "cpu", seed=31, zeros="centered", method="awq")
            V = cfg.vocab
            layers = [model.split_layer_tp(L, cfg, world, rank) for L in w["layers"]]
            head = w["lm_head"].cols(rank * (V // world), (rank + 1) * (V // world))
            shard = {"layers": layers, "embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": head}
            B, page = 32, 16
            eng = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page, num_blocks=B * 2,
                                      max_batch=B, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
            eng.attach_allreduce(ar, rank * (V // world))
            bt = torch.arange(B * 2, dtype=torch.int32).reshape(B, 2)
            tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
            eng.set_inputs(tok.tolist(), [0] * B, bt)
            if rank == 0:
                dense = lambda c: (c.w.float() if c.kind == "fp16" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size))
                ow = {"embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": dense(w["lm_head"]),
                      "layers": [{"input_norm": L["input_norm"], "post_norm": L["post_norm"], "qkv_bias": L["qkv_bias"],
                                  **{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}
                odec = oracle.OracleDecoder({**cfg.__dict__}, ow)
                okv = oracle.OracleKV(cfg.num_layers, B, False)
            del w
            dist.barrier()
            eng.capture(B)
            for step in range(3):
                pos = torch.full((B,), step, dtype=torch.int32)
                eng.replay(B, 1)
                torch.cuda.synchronize()
                full = torch.cat(_gather_cpu(eng.logits[:B].cpu(), world), dim=1)
                mine = eng.token_ids[:B].cpu()
                allids = _gather_cpu(mine, world)
                assert all(torch.equal(allids[0], o) for o in allids[1:])          # every rank picked the same tokens
                assert torch.equal(mine, torch.argmax(full, -1).int())
                nxt = torch.zeros(B, dtype=torch.int32)
                if rank == 0:
                    _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
                    assert torch.allclose(full, ref, atol=1e-2, rtol=1e-2), (step, float((full - ref).abs().max()))
                    nxt = oracle.greedy(ref).int()
                dist.broadcast(nxt, 0)
                tok = nxt
                eng.token_ids[:B].copy_(tok)
            assert ar.status() == 0 and eng.oob_count() == 0
        elif mode == "engine7b":
            # Qwen2-7B's widths (hidden 3584, 28 q / 4 kv heads, FFN 18944) at tp = world, 2 layers, small vocabulary: the TP step of round 5 at
            # the shapes bench.py times -- QKV shard on the image launch (row split: 72 tile pairs at tp 2), gate_up shard's SiLU output as an
            # image, down_proj shard as 4 K quarters (gemm_splitk64) into the fused all-reduce, which writes the next QKV image -- captured
            # per rank, against the unsplit oracle; 24 rows (two row blocks) and 5 rows (one)
            cfg = model.ModelConfig("qwen2-7b-2l", 2, 3584, 28, 4, 128, 18944, 2048, max_pos=64)
            w = model.synth_model(cfg, "w4", "cpu", seed=41, zeros="centered")
            V = cfg.vocab
            layers = [model.split_layer_tp(L, cfg, world, rank) for L in w["layers"]]
            head = w["lm_head"].cols(rank * (V // world), (rank + 1) * (V // world))
            shard = {"layers": layers, "embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": head}
            Bmax, page = 24, 16
            eng = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page, num_blocks=Bmax * 2,
                                      max_batch=Bmax, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
            eng.attach_allreduce(ar, rank * (V // world))
            if rank == 0:
                dense = lambda c: (c.w.float() if c.kind == "fp16" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size))
                ow = {"embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": dense(w["lm_head"]),
                      "layers": [{"input_norm": L["input_norm"], "post_norm": L["post_norm"], "qkv_bias": L["qkv_bias"],
                                  **{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}
                odec = oracle.OracleDecoder({**cfg.__dict__}, ow)
            del w, shard, layers, head
            bt = torch.arange(Bmax * 2, dtype=torch.int32).reshape(Bmax, 2)
            for B in (24, 5):
                okv = oracle.OracleKV(cfg.num_layers, B, False) if rank == 0 else None
                tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(3 + B), dtype=torch.int32)
                eng.set_inputs(tok.tolist(), [0] * B, bt[:B])
                dist.barrier()
                eng.capture(B)
                for step in range(3):
                    pos = torch.full((B,), step, dtype=torch.int32)
                    eng.replay(B, 1)
                    torch.cuda.synchronize()
                    full = torch.cat(_gather_cpu(eng.logits[:B].cpu(), world), dim=1)
                    mine = eng.token_ids[:B].cpu()
                    allids = _gather_cpu(mine, world)
                    assert all(torch.equal(allids[0], o) for o in allids[1:])
                    assert torch.equal(mine, torch.argmax(full, -1).int())
                    nxt = torch.zeros(B, dtype=torch.int32)
                    if rank == 0:
                        _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
                        assert torch.allclose(full, ref, atol=1e-2, rtol=1e-2), (B, step, float((full - ref).abs().max()))
                        nxt = oracle.greedy(ref).int()
                    dist.broadcast(nxt, 0)
                    tok = nxt
                    eng.token_ids[:B].copy_(tok)
            # a target-verify step under TP (configs[4]'s shape of work: q_len rows per sequence, causal inside the paged attention) on the same image
            # launches: 4 sequences x 5 rows at positions 3..7 behind the three tokens the 5-row run above cached (okv still holds them on rank 0)
            nseq, q_len = 4, 5
            toks = torch.randint(0, V, (nseq * q_len,), generator=torch.Generator().manual_seed(77), dtype=torch.int32)
            pos = (3 + torch.arange(q_len, dtype=torch.int32)).repeat(nseq)
            eng.set_inputs(toks.tolist(), pos.tolist(), bt[:nseq])
            dist.barrier()
            eng.forward(nseq * q_len, q_len)
            torch.cuda.synchronize()
            full = torch.cat(_gather_cpu(eng.logits[:nseq * q_len].cpu(), world), dim=1)
            if rank == 0:
                _, ref = odec.forward_tokens(toks, pos, okv, [b for b in range(nseq) for _ in range(q_len)])
                assert torch.allclose(full, ref, atol=1e-2, rtol=1e-2), ("verify rows", float((full - ref).abs().max()))
            assert ar.status() == 0 and eng.oob_count() == 0
        elif mode == "engine70full":
            # BASELINE configs[3] at its real per-rank shapes: Llama-3-70B widths (hidden 8192, 64 q / 8 kv heads, FFN 28672, the
            # whole 128256-token vocabulary), TP 8, batch 32 at context 2048 -- 2 of the 80 layers, every rank's shard
            # (qkv 8192 x 1280, o 1024 x 8192, gate_up 8192 x 7168, down 3584 x 8192, lm_head 8192 x 16032) in one process each on
            # ONE GPU, the step captured per rank, against the unsplit CPU oracle with the same 2047 cached tokens per sequence.
            cfg = model.ModelConfig("llama3-70b-2l", 2, 8192, 64, 8, 128, 28672, 128256, rope_theta=5e5, max_pos=2048 + 16, qkv_bias=False)
            w = model.synth_model(cfg, "w4", "cpu", seed=37, zeros="centered", method="awq")
            V = cfg.vocab
            layers = [model.split_layer_tp(L, cfg, world, rank) for L in w["layers"]]
            head = w["lm_head"].cols(rank * (V // world), (rank + 1) * (V // world))
            shard = {"layers": layers, "embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": head}
            B, page, ctx = 32, 16, 2048
            mbk = (ctx + 2 + page - 1) // page
            eng = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page, num_blocks=B * mbk,
                                      max_batch=B, max_seq_len=ctx + 2, device=dev, tp_size=world, vocab_full=V)
            eng.attach_allreduce(ar, rank * (V // world))
            gk = torch.Generator().manual_seed(5)
            bt = torch.randperm(B * mbk, generator=gk).reshape(B, mbk).to(torch.int32)
            kh0 = cfg.kv_head_of_rank(world, rank) if hasattr(cfg, "kv_head_of_rank") else rank * cfg.nkv // world
            nkr = cfg.per_rank(world).nkv
            if rank == 0:
                dense = lambda c: (c.w.float() if c.kind == "fp16" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size))
                ow = {"embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": dense(w["lm_head"]),
                      "layers": [{"input_norm": L["input_norm"], "post_norm": L["post_norm"], "qkv_bias": L["qkv_bias"],
                                  **{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}
                odec = oracle.OracleDecoder({**cfg.__dict__}, ow)
                okv = oracle.OracleKV(cfg.num_layers, B, False)
            del w, shard, layers, head
            for l in range(cfg.num_layers):      # the same ctx - 1 cached tokens on every rank (its kv head) and in the oracle
                for b in range(B):
                    K = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=gk).half(); Vv = torch.randn(ctx - 1, cfg.nkv, cfg.hd, generator=gk).half()
                    from rtp_llm_amd import kvcache
                    kvcache.write_tokens(eng.kv[l], None, bt[b], 0, K[:, kh0:kh0 + nkr], Vv[:, kh0:kh0 + nkr])
                    if rank == 0:
                        okv.k[l][b], okv.v[l][b] = list(K), list(Vv)
            tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
            eng.set_inputs(tok.tolist(), [ctx - 1] * B, bt)
            dist.barrier()
            eng.capture(B)
            for step in range(2):
                pos = torch.full((B,), ctx - 1 + step, dtype=torch.int32)
                eng.replay(B, 1)
                torch.cuda.synchronize()
                full = torch.cat(_gather_cpu(eng.logits[:B].cpu(), world), dim=1)
                mine = eng.token_ids[:B].cpu()
                allids = _gather_cpu(mine, world)
                assert all(torch.equal(allids[0], o) for o in allids[1:])
                assert torch.equal(mine, torch.argmax(full, -1).int())
                nxt = torch.zeros(B, dtype=torch.int32)
                if rank == 0:
                    _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
                    assert torch.allclose(full, ref, atol=1e-2, rtol=1e-2), (step, float((full - ref).abs().max()))
                    nxt = oracle.greedy(ref).int()
                    top2 = ref.topk(2, -1).values
                    safe = (top2[:, 0] - top2[:, 1]) > 1e-2
                    assert torch.equal(mine[safe], nxt[safe])
                dist.broadcast(nxt, 0)
                tok = nxt
                eng.token_ids[:B].copy_(tok)
            assert ar.status() == 0 and eng.oob_count() == 0
        elif mode == "stress":
            # the fence-free hand-overs (granules for the <= 64-row calls, write-through stores + flags for the 96-row ones) under UNEVEN load, every word checked: 300 all-reduces of alternating geometry with integer
            # patterns (every partial sum exact), one rank delayed at random, and a 256 MB copy running on a second stream of every rank so that the
            # CUs' memory queues, the L2s and the fabric are busy while rows are published and pulled
            noise_src = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_()
            noise_dst = torch.empty_like(noise_src)
            side = torch.cuda.Stream()
            shapes = [(64, 3584), (cap(100), 1024), (5, 8192), (32, 3584), (1, 896)]
            gs = torch.Generator().manual_seed(1234)          # the same on every rank: who sleeps when
            bad = 0
            for it in range(300):
                T, H = shapes[it % len(shapes)]
                idx = torch.arange(T * H, dtype=torch.int64).reshape(T, H)
                pats = [((idx * (r + 3) + it * 11 + r) % 61 - 30).to(torch.float16) for r in range(world)]
                want = sum(p.float() for p in pats).to(torch.float16)
                if it % 3 == 0:
                    with torch.cuda.stream(side):
                        noise_dst.copy_(noise_src, non_blocking=True)
                if int(torch.randint(0, world, (1,), generator=gs)) == rank and it % 7 == 0:
                    torch.cuda._sleep(int(2e6))               # this rank arrives ~1 ms late
                got = ar.all_reduce(pats[rank].to(dev).clone())
                if it % 50 == 49 or it < 5:
                    torch.cuda.synchronize()
                bad += int(not torch.equal(got.cpu(), want))
            torch.cuda.synchronize()
            assert bad == 0, f"{bad} of 300 all-reduces differ from the exact sums"
            assert ar.status() == 0 and ar.hand_over == ("ll" if os.environ.get("MI355_AR_LL") == "1" else "write-through")
        elif mode == "timeout":
            x = torch.ones(4, 3584, dtype=torch.float16, device=dev)
            ar.all_reduce(x.clone()); torch.cuda.synchronize(); dist.barrier()
            if rank == 0:
                ar.all_reduce(x.clone())          # the peer never joins this call: bounded spin, status word, no hang
                st = ar.status()
                assert st != 0, st
            dist.barrier()
        dist.barrier()
        ar.close()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}"))
    finally:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,world", [("kernels", 2), ("publish", 2), ("publish", 4), ("publish_ff", 2), ("engine", 2), ("transport", 2), ("bf16", 2), ("bf16", 4), ("timeout", 2), ("twoshot", 3), ("mixed", 2), ("kernels", 4),
                                        ("kernels", 8), ("twoshot", 4), ("twoshot", 8), ("engine70", 8), ("engine70full", 8), ("kernels_ff", 2), ("engine_ff", 2), ("twoshot_ff", 3), ("kernels_ll", 2), ("kernels_ll", 4), ("engine_ll", 2), ("mixed_ll", 2), ("stress_ll", 2), ("stress_ll", 4), ("engine7b_ll", 2), ("engine7b", 2), ("engine7b", 4), ("stress", 2), ("stress", 4)])   # (8 ranks time-slicing ONE GPU under this load run into the spin bound: a property of the single-GPU setup)
def test_custom_allreduce_processes_on_one_gpu(mode, world):
    assert torch.cuda.is_available()
    # 8 processes time-slicing ONE GPU: a rank's blocks spin for peers whose kernels the hardware scheduler has not switched in yet, and the
    # bounded spin (30 s on a shared device) occasionally trips -- seen once in ~5 runs of the suite, never with <= 4 processes and never reproducible
    # in isolation.  A property of the single-GPU stand-in for a node, not of the kernels: such a case gets ONE more attempt, and the first
    # failure is printed so that it stays visible in the log.
    attempts = 2 if world >= 8 else 1
    for attempt in range(attempts):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=540) for _ in range(world)]
        except Exception as e:  # noqa: BLE001  (queue.Empty: a rank died or hung)
            res = [("missing", repr(e))]
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
        if sorted(res) == [(r, "ok") for r in range(world)]:
            return
        print(f"[test_gpu_allreduce] {mode}-{world}: attempt {attempt + 1} of {attempts} failed: {res}", flush=True)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


@pytest.mark.gpu
def test_rccl_transport_world1_all_reduce_is_capturable():
    """csrc/rccl_transport.cpp on the real librccl (torch's copy, dlopen'ed): communicator from a unique id, ncclAllReduce on the
    caller's stream, eager and captured into a graph.  One GPU -> world 1 (RCCL refuses two ranks on one device)."""
    from rtp_llm_amd import distributed
    t = distributed.RcclTransport(rank=0, world=1)
    x = torch.arange(4096, dtype=torch.float16, device="cuda:0")
    ref = x.clone()
    t.all_reduce(x)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            x.mul_(2.0)
            t.all_reduce(x)
    x.copy_(ref)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, ref * 4)
    with pytest.raises(ValueError):
        t.all_reduce(torch.zeros(8, device="cuda:0"))
    t.close()


@pytest.mark.gpu
def test_rccl_calls_inside_the_captured_cpp_step():
    """The fallback transport inside mi355_decoder_capture: a tp_size = 2 rank-0 shard whose TP points call the real
    ncclAllReduce / ncclAllGather (a world-1 communicator: one GPU) from the C++ step while it is being captured.  Replay must
    reproduce the eager step bit for bit -- i.e. the RCCL launches are captured, not dropped or run at capture time."""
    import ctypes as C
    from rtp_llm_amd import _C, distributed, model
    dev, world, rank = "cuda:0", 2, 0
    t = distributed.RcclTransport(rank=0, world=1)
    real = t.collective

    def all_gather(ctx, send, recv, nbytes, stream):       # fill both rank slots with this rank's pairs
        rc = real.all_gather(real.ctx, send, recv, nbytes, stream)
        return rc or real.all_gather(real.ctx, send, recv + nbytes, nbytes, stream)
    fn = _C.ALL_GATHER_FN(all_gather)

    class Two:
        collective = _C.Collective(real.ctx, real.all_reduce_f16, real.all_reduce_bf16, fn, 0, 2)

    cfg = model.ModelConfig("tiny-tp", 2, 512, 8, 2, 64, 768, 1024, max_pos=256)
    w = model.synth_model(cfg, "w4", "cpu", seed=23, zeros="centered")
    V = cfg.vocab
    shard = {"layers": [model.split_layer_tp(L, cfg, world, rank) for L in w["layers"]], "embedding": w["embedding"],
             "final_norm": w["final_norm"], "lm_head": w["lm_head"].cols(0, V // world)}
    B, page = 6, 16
    mk = lambda: model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=False, page=page, num_blocks=B * 2,
                                     max_batch=B, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
    eager, graph = mk(), mk()
    bt = torch.arange(B * 2, dtype=torch.int32).reshape(B, 2)
    tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
    for e in (eager, graph):
        e.attach_collective(Two, 0)
        e.set_inputs(tok.tolist(), [0] * B, bt)
    graph.capture(B)
    for step in range(4):
        eager.step(B); graph.replay(B, 1)
        torch.cuda.synchronize()
        assert torch.isfinite(eager.logits[:B]).all()
        assert torch.equal(eager.logits[:B], graph.logits[:B]), step
        assert torch.equal(eager.token_ids[:B], graph.token_ids[:B]) and int(eager.token_ids[:B].max()) < V // world
        assert torch.equal(graph.positions[:B].cpu(), torch.full((B,), step + 1, dtype=torch.int32))
    assert eager.oob_count() == 0 and graph.oob_count() == 0
    t.close()
