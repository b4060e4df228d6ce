#!/usr/bin/env python3
"""Decode benchmark — the reference's batch-decode protocol on MI355X.

Mirrors rtp_llm/test/perf_test/batch_decode_test.py + BatchDecodeScheduler
(docs/benchmark/benchmark.md:1-12): exactly `batch` sequences, KV cache allocated and filled
without running prefill, every step decodes one token per sequence (greedy, fed back on
device), W warm-up steps then K timed steps.

    python bench.py [--gpus N --steps K --warmup W] [--workload NAME --batch B --ctx C]

One JSON line on stdout (rank 0): BASELINE.json's metric (decode tokens/s + p50 latency,
Qwen2-7B W4A16, seq 1024), plus `roofline` (dominant kernel = the weight-only dequant GEMM,
algorithmic bytes / HIP-event time) and `cpu_baseline` (the CPU oracle timed on this host).
Data: synthetic (random-init weights of the named architecture, random KV) — there is no
network for checkpoints.  N > 1 (one process per GPU, torchrun env): the headline is the tensor-parallel layout (TP degree
= the largest valid divisor of N) with the hand-written xGMI all-reduce inside the C++ step; N independent
replicas (no data-path collective, weak scaling) are measured first and reported in `replica_layout` -- they also are the
fallback headline if the TP section fails.
"""
import argparse
import ctypes as C
import json
import math
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TRAFFIC_FILE = "r06_traffic.json"   # profiles/: PMC traffic of the engine's linears, tools/engine_traffic.sh

WORKLOADS = {
    # name: (model, weight kind, kv_int8, default batch, default ctx, page)
    "qwen2-7b-w4a16": ("qwen2-7b", "w4", False, 64, 1024, 16),           # BASELINE.json metric (GPTQ g128, fp16 KV)
    "qwen2-7b-w8a16": ("qwen2-7b", "int8", False, 16, 1024, 16),         # configs[1]
    "qwen2-7b-w4a16-kv8": ("qwen2-7b", "w4", True, 64, 4096, 16),        # configs[2]
    "qwen2-7b-w4a16-page64": ("qwen2-7b", "w4", False, 64, 1024, 64),    # the metric's config on the reference's default block size (seq_size_per_block = 64)
    "qwen2-0.5b-fp16": ("qwen2-0.5b", "fp16", False, 1, 128, 16),        # configs[0] shape (GPU run of the plumbing config)
    "llama3-70b-awq": ("llama3-70b", "w4", False, 32, 2048, 16),         # configs[3]: needs --gpus 8 (tp8) or --shard-of 8
    "qwen2-72b-w4a16": ("qwen2-72b", "w4", False, 8, 1024, 16),          # configs[4] target model: --gpus 8 or --shard-of 8
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


TRAFFIC_KERNEL_SOURCES = ("common.h", "gemm_common.h", "gemm.hip", "gemm_wide.hip", "gemm_fullk.h", "gemm_fullk64.hip", "gemm_splitk64.hip",
                          "elementwise.hip")


def gemm_sources_sha():
    """Hash of the sources of the kernels the committed PMC traffic file measured: the four quantised linears of the 5-64-row step are
    gemm_fullk64_kernel (qkv, o), gemm_wide_kernel (gate_up) and gemm_splitk64_kernel (down), and the launch that consumes their slabs /
    writes their input images is add_rmsnorm_kernel (elementwise.hip); gemm.hip holds the entry points and planners.  The few-row
    full-K kernels (gemm_fullk.hip, <= 4 rows) and the prefill GEMM are not launched by that workload, so edits there do not stale the
    file."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rtp_llm_amd", "csrc")
    for f in TRAFFIC_KERNEL_SOURCES:
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def bytes_per_step(cfg, eng, B, ctx, kv_int8):
    """Algorithmic HBM bytes of one decode step: packed linear weights (+scales/zeros) once,
    lm_head once, KV of every cached token once."""
    kv_tok = (2 * cfg.nkv * cfg.hd * (1 if kv_int8 else 2) + (2 * cfg.nkv * 4 if kv_int8 else 0)) * cfg.num_layers
    return {"linears": eng.packed_bytes, "lm_head": eng.packed_bytes_lm_head, "kv": B * ctx * kv_tok}


def fill_kv_random(eng, B, ctx, seed):
    """Perf-mode cache: random K/V for the first ctx-1 tokens of every sequence ("allocate KV without prefill")."""
    g = torch.Generator(device=eng.device).manual_seed(seed)
    for l in range(eng.cfg.num_layers):
        kv = eng.kv[l]
        if kv.dtype == torch.int8:
            kv.copy_(torch.randint(-127, 128, kv.shape, device=eng.device, generator=g, dtype=torch.int8))
            eng.kv_scale[l].copy_(torch.rand(eng.kv_scale[l].shape, device=eng.device, generator=g) * 0.02 + 0.005)
        else:
            kv.copy_(torch.randn(kv.shape, device=eng.device, generator=g, dtype=torch.float16).to(kv.dtype))


def cpu_baseline(cfg, kind, kv_int8, B, ctx, budget_s=25.0):
    """Time the CPU oracle (oracle/oracle.py) on this host: a reduced-layer proxy scaled linearly in L
    (the reference's own perf knob hack_layer_num, docs/benchmark/benchmark.md)."""
    from oracle import oracle
    from rtp_llm_amd import model
    # the oracle works on per-sequence tensors (small ops): beyond ~16 threads torch's intra-op pool only adds overhead
    # (measured on the 256-core GPU host: 61 s per layer at 256 threads vs ~1 s at 8-16)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    nl = 2
    small = model.ModelConfig(cfg.name, nl, cfg.hidden, cfg.nh, cfg.nkv, cfg.hd, cfg.inter, cfg.vocab, cfg.rope_theta,
                              cfg.rms_eps, cfg.qkv_bias, ctx + 8)
    gen = torch.Generator().manual_seed(0)
    dense = lambda c: (c.w.float() if c.kind == "fp16" else oracle.dequant_int8(c.q, c.scales) if c.kind == "int8"
                       else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size))
    layers = []
    for _ in range(nl):
        L = model.synth_layer(small, kind, "cpu", gen)
        layers.append({"input_norm": L["input_norm"], "post_norm": L["post_norm"], "qkv_bias": L["qkv_bias"],
                       **{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}})
    w = {"layers": layers, "embedding": torch.randn(1024, cfg.hidden).half(), "final_norm": torch.ones(cfg.hidden).half(),
         "lm_head": torch.zeros(cfg.hidden, 8)}  # lm_head timed separately below
    dec = oracle.OracleDecoder({**small.__dict__}, w)
    kv = oracle.OracleKV(nl, B, kv_int8)
    # pre-populate ctx-1 cached tokens per sequence (stacked once, not via append)
    for l in range(nl):
        for b in range(B):
            K = torch.randn(ctx - 1, cfg.nkv, cfg.hd).half(); V = torch.randn(ctx - 1, cfg.nkv, cfg.hd).half()
            if kv_int8:
                Kq, ks = oracle.quant_kv_int8(K); Vq, vs = oracle.quant_kv_int8(V)
                kv.k[l][b], kv.v[l][b], kv.ks[l][b], kv.vs[l][b] = list(Kq), list(Vq), list(ks), list(vs)
            else:
                kv.k[l][b], kv.v[l][b] = list(K), list(V)
    ids = torch.randint(0, 1024, (B,), dtype=torch.int32)
    times = []
    t_start = time.time()
    for it in range(4):
        pos = torch.full((B,), ctx - 1 + it, dtype=torch.int32)
        t0 = time.time()
        dec.forward_tokens(ids, pos, kv, list(range(B)))
        times.append(time.time() - t0)
        if time.time() - t_start > budget_s:
            break
    t_layers = statistics.median(times[1:] or times) / nl
    lm = torch.randn(cfg.hidden, cfg.vocab)
    x = torch.randn(B, cfg.hidden).half()
    t0 = time.time(); oracle.greedy(oracle.linear(x, lm, out_f32=True)); t_lm = time.time() - t0
    step = t_layers * cfg.num_layers + t_lm
    return {"value": round(B / step, 2), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/oracle.py OracleDecoder: {nl} of {cfg.num_layers} layers x {len(times)} steps at batch {B} ctx {ctx} "
                      f"(median, first step dropped) + one lm_head, scaled linearly to {cfg.num_layers} layers; weights dequantised once",
            "ms_per_step": round(step * 1e3, 1)}


def spawn_plan(n, argv, port, one_gpu=False):
    """The N child processes `python bench.py --gpus N ...` starts when it is NOT already one rank of a launcher (no WORLD_SIZE in the
    environment): one process per GPU with the torchrun environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT),
    the same command line.  The reference starts one process per rank the same way (rtp_llm/start_server.py:597-720; test idiom
    modules/base/rocm/test/trt_allreduce_test.py:381-470)."""
    plan = []
    for r in range(n):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
               "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0", "MI355_BENCH_SELF_SPAWNED": "1"}
        if one_gpu:
            env["MI355_BENCH_ONE_GPU"] = "1"
        plan.append({"cmd": [sys.executable, os.path.abspath(__file__)] + list(argv), "env": env})
    return plan


def spawn_ranks(n, argv, child_cmd=None, grace_s=90.0):
    """Start the ranks of spawn_plan, wait for all of them.  Rank 0 inherits stdout (it prints the one JSON line); the other ranks' stdout
    goes to stderr.  A rank that exits non-zero makes the whole run non-zero: its peers get `grace_s` to finish (the bench has its own
    fallbacks and watchdogs), then the ones THIS process started are terminated."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    plan = spawn_plan(n, argv, port, one_gpu=os.environ.get("MI355_BENCH_ONE_GPU") == "1")
    procs = []
    for r, p in enumerate(plan):
        cmd = child_cmd if child_cmd is not None else p["cmd"]
        procs.append(subprocess.Popen(cmd, env={**os.environ, **p["env"]}, stdout=None if r == 0 else sys.stderr))
    rc, failed_at = 0, None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0) and rc == 0:
                rc, failed_at = p.returncode, time.time()
                log(f"[bench] rank {r} exited with {p.returncode}")
        if failed_at is not None and time.time() - failed_at > grace_s:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    for r, p in enumerate(procs):
        try:
            p.wait(timeout=10)
        except Exception:  # noqa: BLE001
            p.kill()
        if p.returncode != 0 and rc == 0:
            rc = p.returncode or 1
            log(f"[bench] rank {r} exited with {p.returncode}")
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="qwen2-7b-w4a16", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--ctx", type=int, default=None)
    ap.add_argument("--page", type=int, default=None, help="KV block size in tokens (default: the workload's)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result is then labelled invalid)")
    ap.add_argument("--shard-of", type=int, default=0, help="debug only: run ONE rank's shard of a tp=N layout without the collectives "
                    "(per-rank kernel shapes on a 1-GPU box; result is labelled invalid)")
    ap.add_argument("--no-tp", action="store_true", help="N > 1: skip the tensor-parallel layout measured after the replica headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--debug-set", default="", help="debug only: 'idx=val,...' forwarded to mi355_debug_set (kernel A/B switches)")
    ap.add_argument("--attn-ps", type=int, default=0, help="debug only (tuning build): tokens per attention partition instead of the planner's choice")
    ap.add_argument("--prefetch", type=int, default=None, help="weight-prefetch mask (mi355_decoder_set_weight_prefetch); default: the engine's")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (no launcher): this process becomes the launcher of N ranks
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    if args.debug_set or args.attn_ps:   # kernel A/B switches exist only in the tuning build (python -m rtp_llm_amd.build --tuning)
        os.environ["MI355_TUNING_LIB"] = "1"
    from rtp_llm_amd import _C, distributed, model

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # under a launcher the launcher's world is the truth (torchrun --nproc-per-node); without one, --gpus N > 1 spawned the ranks above
        log(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE")
    if os.environ.get("MI355_BENCH_ONE_GPU") == "1":   # debug: several ranks on device 0 (control-plane check on a 1-GPU box)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _C.lib()  # fail loudly when the HIP extension is missing
    for kv in filter(None, args.debug_set.split(",")):
        _C.lib().mi355_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
    if args.attn_ps:
        _C.lib().mi355_debug_set_attn.argtypes = [C.c_int]
        _C.lib().mi355_debug_set_attn(args.attn_ps)

    mname, kind, kv_int8, dB, dctx, page = WORKLOADS[args.workload]
    page = args.page or page
    B, ctx = args.batch or dB, args.ctx or dctx
    cfg_full = model.MODELS[mname]
    if args.layers:
        cfg_full = model.ModelConfig(**{**cfg_full.__dict__, "num_layers": args.layers})
    total_steps = args.steps + args.warmup + 8 + 40
    cfg_full0, kind0, kv_int8_0, B0, ctx0, page0 = cfg_full, kind, kv_int8, B, ctx, page

    # ---- control plane for N > 1: gloo (CPU tensors) for the barrier and the max-over-ranks; RCCL only carries the
    # data-path collectives of the TP layout below, so the headline does not hang on it
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def build_engine(tp, tp_rank, replica, spec=None, dtype=torch.float16):
        """Synthetic weights generated per rank directly at per-rank shapes (the TP split of random tensors is random
        tensors; norms / embedding use the same seed on every rank).  spec = (cfg_full, kind, kv_int8, B, ctx, page)
        overrides the command-line workload (the extra driver-timed workloads of the default run)."""
        cfg_full, kind, kv_int8, B, ctx, page = spec or (cfg_full0, kind0, kv_int8_0, B0, ctx0, page0)
        max_seq_len = ctx + total_steps + 64
        blocks_per_seq = (max_seq_len + page - 1) // page
        num_blocks = B * blocks_per_seq
        cfg = cfg_full.per_rank(tp)
        if args.shard_of > 1 and world == 1:
            cfg = cfg_full.per_rank(args.shard_of)
        gen = torch.Generator(device=dev).manual_seed(1000 + tp_rank)
        layers = [model.synth_layer(cfg, kind, dev, gen, zeros="centered") for _ in range(cfg.num_layers)]   # zero-mean weights: activations stay O(1)
        gshared = torch.Generator(device=dev).manual_seed(7)
        for L in layers:  # replicated tensors must be identical on all TP ranks
            L["input_norm"] = (1.0 + 0.1 * torch.randn(cfg.hidden, device=dev, generator=gshared)).half()
            L["post_norm"] = (1.0 + 0.1 * torch.randn(cfg.hidden, device=dev, generator=gshared)).half()
        weights = {
            "layers": layers,
            "embedding": (torch.randn(cfg_full.vocab, cfg.hidden, device=dev, generator=gshared) * 0.5).half(),
            "final_norm": (1.0 + 0.1 * torch.randn(cfg.hidden, device=dev, generator=gshared)).half(),
            "lm_head": model.synth_linear(cfg.hidden, cfg.vocab, "fp16", dev, gen),
        }
        shard_dbg = args.shard_of > 1 and world == 1
        eng = model.DecoderEngine(cfg, weights, kv_int8=kv_int8, page=page, num_blocks=num_blocks, max_batch=B,
                                  max_seq_len=max_seq_len, device=dev, tp_size=args.shard_of if shard_dbg else tp, vocab_full=cfg_full.vocab, dtype=dtype)
        if shard_dbg:
            # ONE rank's tensor-parallel step as the TP engine runs it -- per-rank shapes, the fused all-reduce launches in place -- with a
            # world-1 all-reduce context (the rank "exchanges" with itself: same launches and flag protocol, no xGMI hop, sums of one rank:
            # numbers meaningless, timing = the rank's kernels)
            import torch.distributed as dist
            if not dist.is_initialized():
                import socket
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
                dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
            ar1 = distributed.CustomAllReduce(max_bytes=B * cfg_full.hidden * 2, rank=0, world=1)
            eng.attach_allreduce(ar1, 0)
            eng._ar_keep = ar1
        del weights, layers
        torch.cuda.empty_cache()
        fill_kv_random(eng, B, ctx, seed=2 + rank)
        ids0 = torch.randint(0, cfg_full.vocab, (B,), generator=torch.Generator().manual_seed(1 + replica), dtype=torch.int32)
        bt = torch.randperm(num_blocks, generator=torch.Generator().manual_seed(2)).reshape(B, blocks_per_seq).to(torch.int32)
        return cfg, eng, (lambda: eng.set_inputs(ids0.tolist(), [ctx - 1] * B, bt))

    def timed(run, reset):
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
        reset()
        run(args.warmup)
        torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        # p50 step latency (separate pass, per-step events on the launch stream)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(33)]
        evs[0].record()
        for i in range(32):
            run(1)
            evs[i + 1].record()
        torch.cuda.synchronize()
        return elapsed, statistics.median(evs[i].elapsed_time(evs[i + 1]) for i in range(32))

    # ---- headline layout: every GPU is a replica of the whole model with its own `batch` sequences (decode requests
    # are independent units: no data-path collective, weak scaling).  The TP layout is measured after it.
    tp, dp = 1, world
    t0 = time.time()
    cfg, eng, reset = build_engine(1, 0, rank)
    reset()
    torch.cuda.synchronize()
    log(f"[rank {rank}] setup {time.time() - t0:.1f}s: {args.workload} B={B} ctx={ctx} replicas={dp} "
        f"weights {eng.packed_bytes / 1e9:.2f} GB + lm_head {eng.packed_bytes_lm_head / 1e9:.2f} GB")
    graph = None
    if args.prefetch is not None:
        eng.set_weight_prefetch(args.prefetch)
    if not args.no_graph:
        eng.capture(B)
    run = (lambda n: eng.replay(B, n)) if not args.no_graph else (lambda n: [eng.step(B) for _ in range(n)])
    elapsed, p50 = timed(run, reset)
    ms_per_step = elapsed / args.steps * 1e3
    tokens_per_s = B * dp * args.steps / elapsed
    # spread: the K timed steps are ~70 ms of device time at the driver's K = 20; three more blocks of K steps (not part of
    # `value`) show how far one block can sit from the typical one
    repeats = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(args.steps); torch.cuda.synchronize()
        repeats.append(round((time.perf_counter() - t0) / args.steps * 1e3, 4))

    out = {
        "metric": "decode tokens/sec + p50 latency, Qwen2-7B W4A16 b=1..64 @1/2/4/8 GPU" if args.workload == "qwen2-7b-w4a16"
                  else f"decode tokens/sec + p50 latency ({args.workload})",
        "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "p50_ms": round(p50, 4), "ms_per_step_repeats": repeats, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic (random-init weights of the named architecture, random KV cache, greedy decode)",
        "config": {"workload": f"{args.workload}: {mname} decode, weights {kind}"
                               f"{' GPTQ g128' if kind == 'w4' else ''}, KV {'int8' if kv_int8 else 'fp16'}, page {page}",
                   "batch": B, "global_batch": B * dp, "seq_len": ctx, "parallelism": "tp1" if dp == 1 else f"dp{dp} (one full replica per GPU, {B} sequences each)",
                   "graph": not args.no_graph},
    }
    if args.layers:
        out["invalid"] = f"debug run with --layers {args.layers}"
    if args.debug_set or args.attn_ps:
        out["invalid"] = f"debug run with --debug-set {args.debug_set} --attn-ps {args.attn_ps} (tuning build of the library)"
    if args.shard_of > 1:
        out["invalid"] = (f"debug run: ONE rank's step of a tp={args.shard_of} layout on one GPU -- per-rank shapes, the fused all-reduce launches in place "
                          f"with a world-1 context (no xGMI hop)")

    def roofline_of(cfg_r, eng_r, reset_r, label, traffic_ok):
        """`roofline` of the dominant kernel family (the four weight-only dequant GEMMs of a layer) on THIS rank's engine: algorithmic
        bytes per launch (packed weights + activations in and out; under TP the rank's shard of both) / the mean HIP-event
        duration of those launches, measured live on the launch stream by mi355_decoder_profile (eager steps).  Under TP every rank of
        the group must call it together: the eager steps run the in-step all-reduces."""
        bps_r = bytes_per_step(cfg_r, eng_r, B, ctx, kv_int8)
        reset_r()
        prof_r = eng_r.profile(B, 4)
        torch.cuda.synchronize()
        gq = prof_r["gemm_quant"]
        n_launch_step = 4 * cfg_r.num_layers
        act_bytes = B * 2 * (cfg_r.hidden * 2 + cfg_r.nh * cfg_r.hd + (cfg_r.nh + 2 * cfg_r.nkv) * cfg_r.hd + 2 * cfg_r.inter) * cfg_r.num_layers
        alg_bytes_launch = (bps_r["linears"] + act_bytes) / n_launch_step
        avg_ms = gq["ms"] / max(1, gq["launches"])
        ach = alg_bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src, kernel_us = None, None, None
        if traffic_ok:
            try:  # HBM read + write bytes per launch of THESE launches (the engine's four linears per layer), from the committed
                # rocprofv3 --pmc passes over `bench.py --no-graph` (tools/engine_traffic.sh -> profiles/r05_traffic.json); PMC
                # collection needs its own profiled runs, so it cannot happen inside this timed invocation: the value is static --
                # and only quoted while the GEMM sources still hash to what was measured (a stale file reports null)
                tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE))).get(args.workload)
                if tj and tj.get("batch") == B:
                    if tj.get("gemm_sources_sha") == gemm_sources_sha():
                        traffic, traffic_src = int(tj["gemm_quant_bytes_per_launch"]), tj.get("source")
                        kernel_us = tj.get("gemm_quant_kernel_us_per_launch")
                    else:
                        traffic_src = f"stale: profiles/{TRAFFIC_FILE} was collected for different sources of the GEMM / fold kernels (re-run tools/engine_traffic.sh)"
            except Exception:  # noqa: BLE001
                traffic = None
        else:
            traffic_src = "no PMC pass exists for a TP shard (needs the multi-GPU node): null"
        rl = {"bound": "hbm", "kernel": "the four quantised linears of a layer (qkv + RoPE + KV write, o + residual, gate_up + SiLU, down): gemm_fullk64_kernel / "
                                        "gemm_wide_kernel / gemm_splitk64_kernel on the image path, gemm_wq_kernel otherwise", "layout": label,
              "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
              "bytes_per_launch": int(alg_bytes_launch), "avg_launch_us": round(avg_ms * 1e3, 3), "launches_timed": gq["launches"]}
        # `achieved` / `frac` divide by the HIP-event time of EAGER launches, which includes the gap to the next launch (~13 % at b = 64); the same
        # bytes over the kernels' own begin -> end durations (rocprofv3 --kernel-trace of the same eager launches, committed with the traffic file and
        # quoted only while the GEMM sources still hash to what was profiled) are what the per-kernel tables of DESIGN.md use
        if kernel_us:
            rl["avg_kernel_us"] = round(float(kernel_us), 3)
            rl["achieved_kernel_time"] = round(alg_bytes_launch / (float(kernel_us) * 1e-6) / 1e9, 1)
            rl["frac_kernel_time"] = round(rl["achieved_kernel_time"] / HBM_PEAK_GBS, 4)
        return rl, bps_r, prof_r

    if rank == 0 and world > 1:   # N > 1: the line carries `roofline` and `cpu_baseline` too (rank 0's replica here; replaced by its TP shard below)
        out["roofline"], _, _ = roofline_of(cfg, eng, reset, out["config"]["parallelism"], False)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg_full, kind, kv_int8, B, ctx)
    if rank == 0 and world == 1:
        # ---- roofline of the dominant kernel (weight-only dequant GEMM): algorithmic bytes / HIP-event time
        out["roofline"], bps, prof = roofline_of(cfg, eng, reset, "tp1", True)
        total_b = sum(bps.values())
        out["step_roofline"] = {"bytes_per_step": int(total_b), "achieved_GBs": round(total_b / (ms_per_step * 1e-3) / 1e9, 1),
                                "frac_of_8TBs": round(total_b / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "breakdown_bytes": {k: int(v) for k, v in bps.items()},
                                "eager_kernel_ms_per_step": {k: round(v["ms"] / 4, 4) for k, v in prof.items()}}
        # ---- batch sweep of the metric (b = 1..64), same weights / cache
        if not args.no_sweep and not args.no_graph:
            sweep = []
            for b in [x for x in (1, 2, 4, 8, 16, 32, 64) if x <= B]:
                eng.capture(b)
                reset(); eng.replay(b, 4); torch.cuda.synchronize()
                t0 = time.perf_counter(); eng.replay(b, 32); torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 32 * 1e3
                bb = bytes_per_step(cfg, eng, b, ctx, kv_int8)
                sweep.append({"batch": b, "tokens_per_s": round(b / ms * 1e3, 1), "ms_per_step": round(ms, 4),
                              "hbm_frac": round(sum(bb.values()) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            out["sweep"] = sweep
        # ---- the other single-GPU BASELINE configs, timed in this same (driver-run) invocation: configs[2] = W4 + INT8 KV,
        # b=64, ctx 4096; configs[1] = W8A16 (load-time autoquant), b=16, ctx 1024
        if args.workload == "qwen2-7b-w4a16" and not args.no_sweep and not args.no_graph and not args.batch and not args.ctx and not args.layers:
            others = {}
            # ---- prefill of the same weights: 8 prompts x 512 tokens as one chunk pass through the large-M
            # GEMMs + causal paged attention; tokens/s, TFLOP/s (2 x linear params x tokens + attention) and TTFT of the batch
            try:
                npr, plen = min(B, 8), 512          # 4096 rows per launch: the size the large-M kernels were validated at
                pr = [torch.randint(0, cfg_full.vocab, (plen,), generator=torch.Generator().manual_seed(50 + i), dtype=torch.int32).tolist() for i in range(npr)]
                bt_p = torch.arange(npr * ((plen + page - 1) // page), dtype=torch.int32).reshape(npr, -1)
                eng.prefill(pr, bt_p, chunk=plen); torch.cuda.synchronize()
                t0 = time.perf_counter(); eng.prefill(pr, bt_p, chunk=plen); torch.cuda.synchronize()
                tp = time.perf_counter() - t0
                n_lin = cfg.num_layers * (cfg.hidden * (cfg.nh + 2 * cfg.nkv) * cfg.hd + cfg.nh * cfg.hd * cfg.hidden + cfg.hidden * 2 * cfg.inter + cfg.inter * cfg.hidden)
                flops = 2.0 * n_lin * npr * plen + 4.0 * cfg.num_layers * cfg.nh * cfg.hd * npr * (plen * (plen + 1) / 2) + 2.0 * cfg.hidden * cfg.vocab * npr
                others["qwen2-7b-w4a16-prefill"] = {"prompts": npr, "prompt_len": plen, "tokens": npr * plen, "seconds": round(tp, 4),
                                                    "tokens_per_s": round(npr * plen / tp, 1), "tflops": round(flops / tp / 1e12, 1),
                                                    "frac_of_2500_tflops_dense_f16": round(flops / tp / 2.5e15, 4),
                                                    "note": "one chunk pass of 8 x 512 prompt tokens incl. lm_head on the last tokens; TTFT of the batch = seconds"}
            except Exception as e:  # noqa: BLE001
                others["qwen2-7b-w4a16-prefill"] = {"error": f"{type(e).__name__}: {e}"}
            del eng
            for name in ("qwen2-7b-w4a16-kv8", "qwen2-7b-w8a16", "qwen2-7b-w4a16-page64", "qwen2-7b-w4a16-bf16"):
                torch.cuda.empty_cache()
                bf16 = name.endswith("-bf16")       # the metric's config with bf16 activations / KV cache (reference dtype grid)
                mn, kd, k8, b2, c2, pg = WORKLOADS[name[:-5] if bf16 else name]
                cfg2, eng2, reset2 = build_engine(1, 0, 0, (model.MODELS[mn], kd, k8, b2, c2, pg), dtype=torch.bfloat16 if bf16 else torch.float16)
                eng2.capture(b2)
                reset2(); eng2.replay(b2, 4); torch.cuda.synchronize()
                n2 = min(args.steps, 32)
                t0 = time.perf_counter(); eng2.replay(b2, n2); torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / n2 * 1e3
                bb = bytes_per_step(cfg2, eng2, b2, c2, k8)
                others[name] = {"batch": b2, "seq_len": c2, "weights": kd, "kv": "int8" if k8 else ("bf16" if bf16 else "fp16"),
                                "activations": "bf16" if bf16 else "f16", "steps": n2,
                                "tokens_per_s": round(b2 / ms * 1e3, 1), "ms_per_step": round(ms, 4),
                                "bytes_per_step": int(sum(bb.values())),
                                "hbm_frac": round(sum(bb.values()) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                del eng2
            out["other_workloads"] = others
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg_full, kind, kv_int8, B, ctx)
    # ---- N > 1: the metric's multi-GPU layout is tensor parallelism with TP degree = GPU count: Megatron split,
    # the two all-reduce points of every layer run INSIDE the C++ step as fused one-shot peer-read kernels over IPC-mapped
    # buffers (xGMI), greedy sampling as a cross-rank argmax, the whole step replayed as one hipGraph per rank.  That layout
    # becomes the headline; the replica layout measured above is reported beside it.  The collectives were validated with
    # two processes on one GPU only (no multi-GPU box during development), so a failure or hang here falls back to the
    # replica headline instead of costing the line.
    if world > 1 and not args.no_tp:
        import threading
        replica = {"parallelism": out["config"]["parallelism"], "global_batch": B * dp, "tokens_per_s": out["value"],
                   "ms_per_step": out["ms_per_step"], "p50_ms": out["p50_ms"], "ms_per_step_repeats": out["ms_per_step_repeats"], "scaling": "weak"}

        def _tp_timeout():
            if rank == 0:
                print(json.dumps({**out, "tp_layout": {"error": "timed out after 300 s"}}), flush=True)
            os._exit(0)
        watchdog = threading.Timer(300.0, _tp_timeout)
        watchdog.daemon = True
        watchdog.start()
        try:
            import torch.distributed as dist
            ok_tp = lambda t: (world % t == 0 and cfg_full.nh % t == 0 and cfg_full.vocab % t == 0
                               and (cfg_full.nkv % t == 0 or t % cfg_full.nkv == 0))
            tpn = max(t for t in range(1, world + 1) if ok_tp(t))      # Qwen2-7B: 28 q-heads -> tp 1/2/4 (N = 8: tp4 x dp2)
            dpn = world // tpn
            del eng
            torch.cuda.empty_cache()
            grp = None
            for r0 in range(0, world, tpn):   # every rank creates every group (torch.distributed contract); gloo = control plane
                gnew = dist.new_group(list(range(r0, r0 + tpn)), backend="gloo")
                if r0 <= rank < r0 + tpn:
                    grp = gnew
            distributed.set_tp_group(grp)
            tcfg, teng, treset = build_engine(tpn, rank % tpn, rank // tpn)
            # transport of the TP points: the hand-written IPC all-reduce; if the peer mapping is refused on any rank (every
            # rank raises together), RCCL inside the captured step instead -- the reference's own fallback under capture
            # (rocm_rccl.py:511-572) -- so the TP layout survives an IPC failure rather than dropping to replicas
            ar = transport = None
            try:
                if os.environ.get("MI355_BENCH_FORCE_RCCL") == "1":
                    raise RuntimeError("MI355_BENCH_FORCE_RCCL=1")
                ar = distributed.CustomAllReduce(max_bytes=B * cfg_full.hidden * 2, group=grp)
                teng.attach_allreduce(ar, (rank % tpn) * tcfg.vocab)
            except Exception as e:  # noqa: BLE001
                log(f"[rank {rank}] IPC all-reduce unavailable ({type(e).__name__}: {e}); falling back to RCCL inside the step")
                ar = None
                transport = distributed.RcclTransport(group=grp)
                teng.attach_collective(transport, (rank % tpn) * tcfg.vocab)
            if args.prefetch is not None:
                teng.set_weight_prefetch(args.prefetch)
            treset()
            captured = not args.no_graph
            if captured:
                teng.capture(B)
            trun = (lambda n: teng.replay(B, n)) if captured else (lambda n: [teng.step(B) for _ in range(n)])

            def ranks_agree():
                """Every rank of a TP group must hold the SAME bits in the final normed hidden state (rank-order fp32 sums in both all-reduce
                points of every layer): the only check of the cross-device hand-over that exists -- development had one GPU.  A rank that
                read a stale peer row diverges here."""
                treset(); trun(2); torch.cuda.synchronize()
                bits = teng.hidden[:B].contiguous().view(torch.int16).to(torch.int64).cpu()
                t = torch.stack([bits.sum(), (bits * bits).sum(), (bits * torch.arange(1, bits.numel() + 1).view_as(bits) % 1000003).sum()])
                outs = [torch.zeros_like(t) for _ in range(tpn)]
                dist.all_gather(outs, t, group=grp)
                return all(torch.equal(outs[0], o) for o in outs)

            HAND_OVER = {"ll": "data-tagged granules (<= 64-row calls) + write-through publishing stores and flags (opt-in: MI355_AR_LL=1)",
                         "write-through": "write-through publishing stores + drained flags (round 5)",
                         "full-fences": "plain stores + system-scope release / acquire fences (rounds 1-4)"}
            agree = ranks_agree()
            while not agree and ar is not None and ar.hand_over != "full-fences":
                # the fence-free hand-overs have never run across two devices: if the ranks disagree, step down one form (csrc/allreduce.hip,
                # mi355_allreduce_set_protocol), capture again and check again
                nxt = ar.PROTOCOLS[ar.PROTOCOLS.index(ar.hand_over) + 1]
                log(f"[rank {rank}] TP ranks disagree with the {ar.hand_over} hand-over: switching to {nxt}")
                ar.set_protocol(nxt)
                if captured:
                    teng.attach_allreduce(ar, (rank % tpn) * tcfg.vocab)   # drops the captured graphs
                    treset(); teng.capture(B)
                agree = ranks_agree()
            protocol = "rccl" if ar is None else HAND_OVER[ar.hand_over]
            t_el, t_p50 = timed(trun, treset)
            # the same three extra blocks of K steps as for the replica layout, so that `ms_per_step_repeats` of the line belongs to the
            # layout `value` / `ms_per_step` are quoted on (max over ranks: a TP step ends when its slowest rank does)
            t_repeats = []
            for _ in range(3):
                torch.cuda.synchronize(); barrier(); t0r = time.perf_counter(); trun(args.steps); torch.cuda.synchronize()
                t_repeats.append(round(max_over_ranks(time.perf_counter() - t0r) / args.steps * 1e3, 4))
            # roofline of the headline layout: every rank profiles its shard together (the eager steps run the in-step all-reduces)
            tp_roof = None
            try:
                tp_roof, _, _ = roofline_of(tcfg, teng, treset, f"tp{tpn} (rank 0's shard)", False)
            except Exception as e:  # noqa: BLE001
                log(f"[rank {rank}] roofline pass of the TP layout failed ({type(e).__name__}: {e}); the line keeps the replica's")
            st = ar.status() if ar is not None else 0
            if st != 0:
                raise RuntimeError(f"all-reduce spin timed out (status {st})")
            tp_info = {"parallelism": f"tp{tpn}" + (f" x dp{dpn}" if dpn > 1 else ""), "global_batch": B * dpn,
                       "tokens_per_s": round(B * dpn * args.steps / t_el, 1), "ms_per_step": round(t_el / args.steps * 1e3, 4),
                       "p50_ms": round(t_p50, 4), "graph": captured, "ranks_bit_identical": bool(agree), "hand_over": protocol,
                       "collectives": ("hand-written one-shot peer-read all-reduce over IPC/xGMI, fused with split-K reduce + residual + "
                                       "RMSNorm (2 per layer) + cross-rank greedy argmax; no RCCL on the data path") if ar is not None else
                                      ("RCCL fallback (IPC peer mapping unavailable): ncclAllReduce on the local split-K fold (2 per layer) + "
                                       "ncclAllGather of one (max, index) pair per row, all inside the captured step")}
            # headline := the TP layout
            out.update(value=tp_info["tokens_per_s"], ms_per_step=tp_info["ms_per_step"], p50_ms=tp_info["p50_ms"], ms_per_step_repeats=t_repeats,
                       scaling="strong" if dpn == 1 else "strong within a tp group, weak across the dp groups")
            out["config"].update(parallelism=tp_info["parallelism"], global_batch=B * dpn)
            out["tp_layout"], out["replica_layout"] = tp_info, replica
            if tp_roof is not None:
                out["replica_layout"]["roofline"] = out.get("roofline")
                out["roofline"] = tp_roof
        except Exception as e:  # noqa: BLE001
            out["tp_layout"] = {"error": f"{type(e).__name__}: {e}"}
            out["replica_layout"] = replica
            log(f"[rank {rank}] TP layout failed, headline stays on the replica layout: {out['tp_layout']['error']}")
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        # the line is out: a rank that lost its peers (a TP-leg failure on one rank only) must not sit in the collective's
        # own timeout
        import threading
        bye = threading.Timer(60.0, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
        try:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        except Exception as e:  # noqa: BLE001
            log(f"[rank {rank}] shutdown barrier: {type(e).__name__}: {e}")
        bye.cancel()


if __name__ == "__main__":
    main()
