"""Synthetic HF checkpoints for the loader tests, written with *independent* packers (the on-disk formats of AutoGPTQ /
AutoAWQ as the reference loader reads them, model_loader/group_wise_quant_weight.py:35-301, device_impl.py:148-171)."""
import json
import os

import torch
from safetensors.torch import save_file

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


def write_ckpt(tmp, kind, cfg, canon, bf16_aux=False, extra_cfg=None):
    """bf16_aux: store norms / embedding / lm_head / biases as bf16 (the dtype Qwen2 checkpoints ship them in)."""
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
    if bf16_aux:
        for k in list(t):
            if t[k].dtype == torch.float16 and not k.endswith(".scales"):
                t[k] = t[k].to(torch.bfloat16)
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
    hf.update(extra_cfg or {})
    json.dump(hf, open(os.path.join(tmp, "config.json"), "w"))


