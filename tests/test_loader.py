"""Checkpoint loader (HF safetensors -> canonical -> TP split) on synthetic fp16 / GPTQ / AWQ checkpoints written
with *independent* packers (the on-disk formats of AutoGPTQ / AutoAWQ as the reference loader reads them,
model_loader/group_wise_quant_weight.py:35-301, device_impl.py:148-171).  CPU only."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

from oracle import oracle
from rtp_llm_amd import loader, model

AWQ_PACK_ORDER = [0, 2, 4, 6, 1, 3, 5, 7]   # nibble j of a word holds logical column 8c + AWQ_PACK_ORDER[j]


def _pack_rows8(q):        # [K, N] codes -> int32 [K/8, N], 8 consecutive k per word, low nibble first (GPTQ qweight)
    K, N = q.shape
    w = torch.zeros(K // 8, N, dtype=torch.int64)
    for j in range(8):
        w |= q[j::8].to(torch.int64) << (4 * j)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def _pack_cols8(q, order):  # [R, N] codes -> int32 [R, N/8]; nibble j = column 8c + order[j]
    R, N = q.shape
    w = torch.zeros(R, N // 8, dtype=torch.int64)
    qq = q.reshape(R, N // 8, 8).to(torch.int64)
    for j in range(8):
        w |= qq[:, :, order[j]] << (4 * j)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def _write_ckpt(tmp, kind, cfg, canon):
    t = {}

    def put_linear(name, c):
        if kind == "fp16":
            t[name + ".weight"] = c.w.t().contiguous()
        elif kind == "gptq":
            t[name + ".qweight"] = _pack_rows8(c.q)
            t[name + ".qzeros"] = _pack_cols8((c.z_eff.to(torch.int16) - 1).to(torch.uint8), list(range(8)))
            t[name + ".scales"] = c.scales
            t[name + ".g_idx"] = (torch.arange(c.K) // c.group_size).to(torch.int32)
        else:
            t[name + ".qweight"] = _pack_cols8(c.q, AWQ_PACK_ORDER)
            t[name + ".qzeros"] = _pack_cols8(c.z_eff, AWQ_PACK_ORDER)
            t[name + ".scales"] = c.scales
    hd, nh, nkv, I = cfg.hd, cfg.nh, cfg.nkv, cfg.inter
    for i, L in enumerate(canon["layers"]):
        p = f"model.layers.{i}."
        qkv, gu = L["qkv"], L["gate_up"]
        put_linear(p + "self_attn.q_proj", qkv.cols(0, nh * hd))
        put_linear(p + "self_attn.k_proj", qkv.cols(nh * hd, (nh + nkv) * hd))
        put_linear(p + "self_attn.v_proj", qkv.cols((nh + nkv) * hd, (nh + 2 * nkv) * hd))
        b = L["qkv_bias"]
        t[p + "self_attn.q_proj.bias"], t[p + "self_attn.k_proj.bias"], t[p + "self_attn.v_proj.bias"] = \
            b[: nh * hd].clone(), b[nh * hd:(nh + nkv) * hd].clone(), b[(nh + nkv) * hd:].clone()
        put_linear(p + "self_attn.o_proj", L["o"])
        put_linear(p + "mlp.gate_proj", gu.cols(0, I)); put_linear(p + "mlp.up_proj", gu.cols(I, 2 * I))
        put_linear(p + "mlp.down_proj", L["down"])
        t[p + "input_layernorm.weight"], t[p + "post_attention_layernorm.weight"] = L["input_norm"], L["post_norm"]
    t["model.embed_tokens.weight"], t["model.norm.weight"] = canon["embedding"], canon["final_norm"]
    t["lm_head.weight"] = canon["lm_head"].w.t().contiguous()
    keys = sorted(t)
    half = len(keys) // 2                      # two shards + index, like real checkpoints
    save_file({k: t[k].contiguous() for k in keys[:half]}, os.path.join(tmp, "model-00001-of-00002.safetensors"))
    save_file({k: t[k].contiguous() for k in keys[half:]}, os.path.join(tmp, "model-00002-of-00002.safetensors"))
    json.dump({"weight_map": {k: ("model-00001-of-00002.safetensors" if i < half else "model-00002-of-00002.safetensors")
                              for i, k in enumerate(keys)}}, open(os.path.join(tmp, "model.safetensors.index.json"), "w"))
    hf = {"hidden_size": cfg.hidden, "num_hidden_layers": cfg.num_layers, "num_attention_heads": nh, "num_key_value_heads": nkv,
          "intermediate_size": I, "vocab_size": cfg.vocab, "rope_theta": cfg.rope_theta, "rms_norm_eps": cfg.rms_eps,
          "max_position_embeddings": cfg.max_pos, "head_dim": hd}
    if kind != "fp16":
        hf["quantization_config"] = {"quant_method": kind, "bits": 4, "group_size": 128, "desc_act": False}
    json.dump(hf, open(os.path.join(tmp, "config.json"), "w"))


CFG = model.ModelConfig("tiny", 2, 256, 8, 4, 64, 512, 512, rope_theta=1e6, max_pos=128)


@pytest.mark.parametrize("kind", ["fp16", "gptq", "awq"])
def test_load_roundtrip(tmp_path, kind):
    canon = model.synth_model(CFG, "fp16" if kind == "fp16" else "w4", "cpu", seed=4, method="gptq" if kind == "gptq" else "awq")
    _write_ckpt(str(tmp_path), kind, CFG, canon)
    mc, w = loader.load_hf_checkpoint(str(tmp_path))
    assert (mc.num_layers, mc.hidden, mc.nh, mc.nkv, mc.hd, mc.inter, mc.vocab, mc.qkv_bias) == (2, 256, 8, 4, 64, 512, 512, True)
    for L0, L1 in zip(canon["layers"], w["layers"]):
        for k in ("qkv", "o", "gate_up", "down"):
            a, b = L0[k], L1[k]
            assert (a.kind, a.K, a.N) == (b.kind, b.K, b.N)
            for f in ("w", "q", "scales", "z_eff"):
                assert (getattr(a, f) is None) == (getattr(b, f) is None)
                if getattr(a, f) is not None:
                    assert torch.equal(getattr(a, f), getattr(b, f)), (k, f)
        assert torch.equal(L0["qkv_bias"], L1["qkv_bias"]) and torch.equal(L0["input_norm"], L1["input_norm"])
    assert torch.equal(w["lm_head"].w, canon["lm_head"].w) and torch.equal(w["embedding"], canon["embedding"])


def test_load_int8_autoquant_and_tp_split(tmp_path):
    canon = model.synth_model(CFG, "fp16", "cpu", seed=5)
    _write_ckpt(str(tmp_path), "fp16", CFG, canon)
    mc, w = loader.load_hf_checkpoint(str(tmp_path), quantization="int8")
    L = w["layers"][0]
    assert L["qkv"].kind == "int8" and L["qkv"].q.dtype == torch.int8 and w["lm_head"].kind == "fp16"
    # weight-only int8 reproduces the fp16 layer to quantisation accuracy
    x = torch.randn(4, 256).half()
    y8 = oracle.linear(x, oracle.dequant_int8(L["down"].q[:256] if False else L["qkv"].q, L["qkv"].scales))
    assert torch.allclose(y8.float(), oracle.linear(x, canon["layers"][0]["qkv"].w.float()).float(), atol=3e-2, rtol=3e-2)
    # TP=2 per-rank shapes (GPTQ checkpoint): group-aligned row split, vocab-split lm_head
    canon4 = model.synth_model(CFG, "w4", "cpu", seed=6)
    d2 = tmp_path / "gptq"; d2.mkdir()
    _write_ckpt(str(d2), "gptq", CFG, canon4)
    for r in range(2):
        mcr, wr = loader.load_hf_checkpoint(str(d2), tp=2, rank=r)
        assert (mcr.nh, mcr.nkv, mcr.inter, mcr.vocab) == (4, 2, 256, 256)
        Lr = wr["layers"][1]
        assert Lr["qkv"].N == (4 + 2 * 2) * 64 and Lr["o"].K == 256 and Lr["down"].K == 256 and Lr["gate_up"].N == 512
        assert Lr["down"].scales.shape == (2, 256) and wr["lm_head"].N == 256


def test_act_order_checkpoint_is_rejected(tmp_path):
    canon = model.synth_model(CFG, "w4", "cpu", seed=7)
    _write_ckpt(str(tmp_path), "gptq", CFG, canon)
    cfg = json.load(open(tmp_path / "config.json")); cfg["quantization_config"]["desc_act"] = True
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    with pytest.raises(NotImplementedError):
        loader.load_hf_checkpoint(str(tmp_path))
