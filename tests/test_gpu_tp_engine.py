"""The C++ step driver's tensor-parallel segmentation on real kernels: two ranks (both on cuda:0, collectives over gloo,
which stages CUDA tensors through the host) run DecoderEngine.step_tp on the Megatron split of one model and must
reproduce the unsplit oracle.  This is the data flow bench.py's `tp_layout` uses over RCCL."""
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


def _worker(rank, world, port, q, kv_int8):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import torch.distributed as dist
        from oracle import oracle
        from rtp_llm_amd import distributed, model
        dev = "cuda:0"
        distributed.init_distributed("gloo")
        # inter = 768 is not a multiple of tp * 128: exercises the align_size padding too
        cfg = model.ModelConfig("tiny-tp", 2, 512, 8, 2, 64, 768, 1024, max_pos=256)
        w = model.synth_model(cfg, "w4", "cpu", seed=21, zeros="centered")     # same full weights on every rank
        V = cfg.vocab
        layers = [model.split_layer_tp(L, cfg, world, rank) for L in w["layers"]]
        head = w["lm_head"].cols(rank * (V // world), (rank + 1) * (V // world))   # vocab-split lm_head
        shard = {"layers": layers, "embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": head}
        B, page = 5, 16
        eng = model.DecoderEngine(cfg.per_rank(world), model.weights_to(shard, dev), kv_int8=kv_int8, page=page, num_blocks=B * 2,
                                  max_batch=B, max_seq_len=32, device=dev, tp_size=world, vocab_full=V)
        dense = lambda c: (c.w.float() if c.kind == "fp16" else oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size))
        ow = {"embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": dense(w["lm_head"]),
              "layers": [{"input_norm": L["input_norm"], "post_norm": L["post_norm"], "qkv_bias": L["qkv_bias"],
                          **{k: dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}
        odec = oracle.OracleDecoder({**cfg.__dict__}, ow)
        okv = oracle.OracleKV(cfg.num_layers, B, kv_int8)
        bt = torch.arange(B * 2, dtype=torch.int32).reshape(B, 2)
        tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
        eng.set_inputs(tok.tolist(), [0] * B, bt)
        tol = dict(atol=1e-2, rtol=1e-2)                                         # north_star, fp16 and INT8 KV alike
        for step in range(4):
            pos = torch.full((B,), step, dtype=torch.int32)
            _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
            eng.step_tp(B, sample=True)
            torch.cuda.synchronize()
            full = distributed.all_gather(eng.logits[:B], distributed.Group.TP).cpu()
            assert torch.allclose(full, ref, **tol), (step, (full - ref).abs().max())
            assert torch.equal(eng.positions[:B].cpu(), pos + 1)
            # every rank must hold the same greedy tokens (they feed the next step on each rank)
            mine = eng.token_ids[:B].clone()
            g = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(g, mine)
            assert all(torch.equal(g[0], x) for x in g)
            tok = oracle.greedy(ref)
            eng.token_ids[:B].copy_(tok)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()[-800:]}"))
    finally:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kv_int8", [False, True])
def test_engine_step_tp_two_ranks_one_gpu(kv_int8):
    assert torch.cuda.is_available()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kv_int8)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
