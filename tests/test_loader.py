"""Checkpoint loader (HF safetensors -> canonical -> TP split) on synthetic fp16 / GPTQ / AWQ checkpoints written
with *independent* packers (the on-disk formats of AutoGPTQ / AutoAWQ as the reference loader reads them,
model_loader/group_wise_quant_weight.py:35-301, device_impl.py:148-171).  CPU only."""
import json

import pytest
import torch

from oracle import oracle
from rtp_llm_amd import loader, model

from ckpt_util import write_ckpt as _write_ckpt


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
        # the reference's TP layout of the embedding table (hidden split, utils/model_weight.py:1490): the rank's columns of the same table
        _, ws = loader.load_hf_checkpoint(str(d2), tp=2, rank=r, split_embedding=True)
        H = wr["embedding"].shape[1]
        assert ws["embedding"].shape == (wr["embedding"].shape[0], H // 2) and ws["embedding"].is_contiguous()
        assert torch.equal(ws["embedding"], wr["embedding"][:, r * (H // 2):(r + 1) * (H // 2)])


def test_act_order_checkpoint_is_rejected(tmp_path):
    canon = model.synth_model(CFG, "w4", "cpu", seed=7)
    _write_ckpt(str(tmp_path), "gptq", CFG, canon)
    cfg = json.load(open(tmp_path / "config.json")); cfg["quantization_config"]["desc_act"] = True
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    with pytest.raises(NotImplementedError):
        loader.load_hf_checkpoint(str(tmp_path))


def test_bf16_checkpoint_tensors_are_converted(tmp_path):
    """Qwen2 checkpoints ship bf16 norms / embeddings: values that are fp16-exact must load bit-identically."""
    canon = model.synth_model(CFG, "w4", "cpu", seed=8)
    # make the fp16 tensors bf16-representable so that the round trip is exact
    rt = lambda t: t.to(torch.bfloat16).to(torch.float16)
    canon["embedding"], canon["final_norm"] = rt(canon["embedding"]), rt(canon["final_norm"])
    canon["lm_head"].w = rt(canon["lm_head"].w)
    for L in canon["layers"]:
        L["input_norm"], L["post_norm"], L["qkv_bias"] = rt(L["input_norm"]), rt(L["post_norm"]), rt(L["qkv_bias"])
    _write_ckpt(str(tmp_path), "gptq", CFG, canon, bf16_aux=True)
    mc, w = loader.load_hf_checkpoint(str(tmp_path))
    assert w["embedding"].dtype == torch.float16 and torch.equal(w["embedding"], canon["embedding"])
    assert torch.equal(w["lm_head"].w, canon["lm_head"].w) and torch.equal(w["final_norm"], canon["final_norm"])
    for L0, L1 in zip(canon["layers"], w["layers"]):
        assert L1["input_norm"].dtype == torch.float16 and torch.equal(L0["input_norm"], L1["input_norm"])
        assert torch.equal(L0["qkv_bias"], L1["qkv_bias"]) and torch.equal(L0["qkv"].q, L1["qkv"].q)


@pytest.mark.parametrize("extra", [{"rope_scaling": {"rope_type": "mrope", "factor": 8.0}}, {"rope_scaling": {"type": "longrope", "factor": 2.0}},
                                   {"use_sliding_window": True}])
def test_unsupported_position_schemes_are_rejected(tmp_path, extra):
    canon = model.synth_model(CFG, "fp16", "cpu", seed=9)
    _write_ckpt(str(tmp_path), "fp16", CFG, canon, extra_cfg=extra)
    with pytest.raises(NotImplementedError):
        loader.load_hf_checkpoint(str(tmp_path))


@pytest.mark.parametrize("rs", [{"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 64},
                                {"type": "yarn", "factor": 4.0, "original_max_position_embeddings": 64}, {"type": "linear", "factor": 2.0}])
def test_scaled_rope_styles_load_into_the_model_config(tmp_path, rs):
    """rope_scaling of the tabulable styles (models/llama.py:87-117) reaches ModelConfig and changes the rotation table."""
    canon = model.synth_model(CFG, "fp16", "cpu", seed=9)
    _write_ckpt(str(tmp_path), "fp16", CFG, canon, extra_cfg={"rope_scaling": rs})
    mc, _ = loader.load_hf_checkpoint(str(tmp_path))
    assert mc.rope_scaling == rs
    base = model.ModelConfig(**{**mc.__dict__, "rope_scaling": None})
    assert not torch.equal(model.rope_table(mc, "cpu"), model.rope_table(base, "cpu"))


def test_padded_vocab_follows_the_tensor(tmp_path):
    canon = model.synth_model(CFG, "fp16", "cpu", seed=10)
    _write_ckpt(str(tmp_path), "fp16", CFG, canon, extra_cfg={"vocab_size": 500})   # tensors have 512 rows
    mc, w = loader.load_hf_checkpoint(str(tmp_path))
    assert mc.vocab == 512 and w["lm_head"].N == 512


def test_bf16_checkpoint_loads_as_bf16_when_asked(tmp_path):
    """dtype=torch.bfloat16: 16-bit tensors of the checkpoint stay / become bf16 (no fp16 range check), W4 codes and their fp16
    scales are untouched -- the weight dict DecoderEngine(dtype=torch.bfloat16) takes."""
    canon = model.synth_model(CFG, "w4", "cpu", seed=8)
    canon["embedding"][0, 0] = 60000.0                                      # fine in fp16 here, but make one value bf16-only below
    _write_ckpt(str(tmp_path), "gptq", CFG, canon, bf16_aux=True)
    mc, w = loader.load_hf_checkpoint(str(tmp_path), dtype=torch.bfloat16)
    assert w["embedding"].dtype == torch.bfloat16 and w["final_norm"].dtype == torch.bfloat16
    assert w["lm_head"].w.dtype == torch.bfloat16 and w["layers"][0]["qkv_bias"].dtype == torch.bfloat16
    assert w["layers"][0]["qkv"].scales.dtype == torch.float16 and torch.equal(w["layers"][0]["qkv"].q, canon["layers"][0]["qkv"].q)
    assert torch.equal(w["embedding"], canon["embedding"].to(torch.bfloat16))
    with pytest.raises(NotImplementedError):
        loader.load_hf_checkpoint(str(tmp_path), dtype=torch.bfloat16, quantization="int8")
