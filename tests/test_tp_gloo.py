"""Tensor-parallel path on CPU: two processes over gloo (127.0.0.1).  Exercises the collective wrappers the
decode step uses between its segments (rtp_llm_amd.distributed.all_reduce / all_gather, the shape of the
reference's collective_torch.all_reduce(Group.TP), collective_torch.py:694-769) and checks that the Megatron
split of a quantised layer (model.split_layer_tp) reproduces the unsplit oracle layer when the per-rank partial
outputs are summed by all_reduce — the exact data flow of DecoderEngine.step_tp."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, inter, nkv=4):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle
    from rtp_llm_amd import distributed, model
    distributed.init_distributed("gloo")
    try:
        # ---- collectives: closed-form sums (the reference's distributed test style)
        t = torch.full((4, 8), float(rank + 1))
        out = distributed.all_reduce(t.clone(), distributed.Group.TP)
        assert torch.equal(out, torch.full((4, 8), float(sum(range(1, world + 1)))))
        g = distributed.all_gather(torch.full((2, 3), float(rank)), distributed.Group.TP)
        assert g.shape == (2, 3 * world) and all(torch.equal(g[:, 3 * r:3 * r + 3], torch.full((2, 3), float(r))) for r in range(world))
        assert distributed.tp_size() == world and distributed.tp_rank() == rank
        # ---- one TP decoder layer: split weights per rank, all_reduce after O-proj and down-proj.  inter = 384 is not
        # a multiple of tp * 128: the split pads it to 512 with zero weights (the reference's align_size = tp * g)
        # nkv = 1 < tp: the kv head is replicated on both ranks (get_sp_tensor, utils/model_weight.py:447-466)
        cfg = model.ModelConfig("t", 1, 256, 8 if nkv == 4 else 4, nkv, 64 if nkv == 4 else 64, inter, 64, max_pos=64)
        if nkv == 1:
            cfg = model.ModelConfig("t", 1, 256, 4, 1, 64, inter, 64, max_pos=64)
            assert cfg.per_rank(world).nkv == 1 and cfg.per_rank(world).nh == 2
        assert cfg.per_rank(world).inter == (256 if inter == 384 else inter // world)
        gen = torch.Generator().manual_seed(0)                      # same full weights on every rank
        L = model.synth_layer(cfg, "w4", "cpu", gen)
        dense = lambda c: oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size)
        x = (torch.randn(3, 256, generator=gen) * 0.5).half()
        pos = torch.tensor([0, 5, 9])
        cs = oracle.rope_cos_sin(cfg.hd, cfg.rope_theta, cfg.max_pos)

        def layer(Lw, c, reduce):
            xn = oracle.rmsnorm(x, Lw["input_norm"], cfg.rms_eps)
            qkv = oracle.linear(xn, dense(Lw["qkv"]), Lw["qkv_bias"])
            nh, nkv, hd = c.nh, c.nkv, c.hd
            qh = oracle.apply_rope(qkv[:, :nh * hd].reshape(3, nh, hd), pos, cs)
            kh = oracle.apply_rope(qkv[:, nh * hd:(nh + nkv) * hd].reshape(3, nkv, hd), pos, cs)
            vh = qkv[:, (nh + nkv) * hd:].reshape(3, nkv, hd)
            attn = torch.stack([oracle.attention_decode(qh[t], kh[t:t + 1], vh[t:t + 1], hd ** -0.5).reshape(-1) for t in range(3)])
            h = x + reduce(oracle.linear(attn, dense(Lw["o"])))
            xn2 = oracle.rmsnorm(h, Lw["post_norm"], cfg.rms_eps)
            return h + reduce(oracle.linear(oracle.silu_mul(oracle.linear(xn2, dense(Lw["gate_up"]))), dense(Lw["down"])))

        full = layer(L, cfg, lambda t: t)
        mine = layer(model.split_layer_tp(L, cfg, world, rank), cfg.per_rank(world),
                     lambda t: distributed.all_reduce(t.clone(), distributed.Group.TP))
        assert torch.allclose(mine.float(), full.float(), atol=2e-2, rtol=2e-2), (mine.float() - full.float()).abs().max()
        # every rank holds bit-identical results after the all-reduce (required for ranks to agree on greedy tokens)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert all(torch.equal(gathered[0], gi) for gi in gathered)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("inter,nkv", [(512, 4), (384, 4), (512, 1)])
def test_tp2_gloo_layer_and_collectives(inter, nkv):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, inter, nkv)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
