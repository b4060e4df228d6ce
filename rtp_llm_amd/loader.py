"""HF checkpoint -> canonical weights -> (TP split) -> MI355-native packed weights.

The decode-path slice of the reference's loader stack:
  * HF names -> W-names: rtp_llm/models/qwen_v2.py:43-335 (Qwen2), rtp_llm/models/llama_weight.py (Llama);
  * QKV merge [q | k | v] along the output dim (merge_qkv_hf, rtp_llm/utils/model_weight.py) and gate/up merge
    into ffn_w13 (model_loader/group_wise_quant_weight.py:211-260);
  * GPTQ / AWQ tensors `.qweight/.qzeros/.scales` (model_loader/group_wise_quant_weight.py:35-301), canonicalised
    by rtp_llm_amd.quant (device_impl.py:148-171,242-300); like the reference loader, `g_idx` (act-order) is not
    supported — the reference never reads it — and a non-trivial g_idx is rejected here;
  * `--quantization int8`: load-time per-channel autoquant of the fp16 linears (weight_only_quant_weight.py:94-105).

With quantization="int8" the per-channel scales are computed on the full [K, N] weight BEFORE the row-parallel TP split
(the reference autoquantises each rank's shard; both are valid per-channel quantisations, this one makes every rank's
codes a slice of the tp = 1 codes).  bf16 / fp32 checkpoint tensors are converted to fp16 at load time.

Pure host code (safetensors + torch on CPU); the result feeds model.DecoderEngine / Qwen2DecoderModel unchanged.
"""
import json
import os
from typing import Dict, Optional, Tuple

import torch

from . import quant
from .model import CanonLinear, ModelConfig, split_layer_tp


class _Shards:
    """Tensor lookup over one or several safetensors files (model.safetensors.index.json aware)."""

    def __init__(self, path: str):
        from safetensors import safe_open
        self.path = path
        idx = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(idx):
            self.map = json.load(open(idx))["weight_map"]
        else:
            files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
            if not files:
                raise FileNotFoundError(f"no .safetensors files under {path}")
            self.map = {}
            for f in files:
                with safe_open(os.path.join(path, f), "pt") as sf:
                    for k in sf.keys():
                        self.map[k] = f
        self._open = {}
        self._safe_open = safe_open

    def has(self, name: str) -> bool:
        return name in self.map

    def get(self, name: str) -> torch.Tensor:
        f = self.map[name]
        if f not in self._open:
            self._open[f] = self._safe_open(os.path.join(self.path, f), "pt")
        return self._open[f].get_tensor(name)


def config_from_hf(cfg: dict, name: str = "hf-model") -> Tuple[ModelConfig, dict]:
    """config.json -> (ModelConfig, quantization dict) — the fields models/qwen_v2.py:338-399 reads."""
    nh = cfg["num_attention_heads"]
    mc = ModelConfig(name=name, num_layers=cfg["num_hidden_layers"], hidden=cfg["hidden_size"], nh=nh,
                     nkv=cfg.get("num_key_value_heads", nh), hd=cfg.get("head_dim", cfg["hidden_size"] // nh),
                     inter=cfg["intermediate_size"], vocab=cfg["vocab_size"], rope_theta=float(cfg.get("rope_theta", 10000.0)),
                     rms_eps=float(cfg.get("rms_norm_eps", 1e-6)), qkv_bias=False,
                     max_pos=int(cfg.get("max_position_embeddings", 8192)))
    rs = cfg.get("rope_scaling") or {}
    kind = rs.get("rope_type", rs.get("type", "default")) if rs else "default"
    if kind not in ("default", None):
        # linear / llama3 / yarn fold into the position-indexed cos/sin table (model.rope_frequencies); "dynamic" (models/llama.py:95-96,
        # gpt_neox.py:128-134: RopeStyle::DynamicNTK, scale = factor, original context = max_position_embeddings) gets one base per position
        if kind not in ("linear", "llama3", "yarn", "dynamic"):
            raise NotImplementedError(f"rope_scaling={rs}: only base / linear / llama3 / yarn / dynamic RoPE are on this decode path")
        mc.rope_scaling = dict(rs)
        if kind == "dynamic":
            mc.rope_scaling.setdefault("original_max_position_embeddings", int(cfg.get("max_position_embeddings", 2048)))
    if cfg.get("use_dynamic_ntk"):   # Qwen-1 (models/qwen.py:293-295): RopeStyle::QwenDynamicNTK over seq_length
        mc.rope_scaling = {"rope_type": "qwen_dynamic", "original_max_position_embeddings": int(cfg.get("seq_length", 8192))}
    if cfg.get("use_sliding_window"):
        raise NotImplementedError("sliding-window attention is not on this decode path")
    q = cfg.get("quantization_config") or {}
    qc = {"method": q.get("quant_method", "none"), "bits": int(q.get("bits", 16)), "group_size": int(q.get("group_size", 0)),
          "desc_act": bool(q.get("desc_act", False))}
    return mc, qc


def _f16(t: torch.Tensor, name: str, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """Checkpoint tensor -> the activation dtype of the model.  fp16 (default): bf16 / fp32 checkpoints (Qwen2 ships bf16) are
    converted once at load time; a value outside the fp16 range would become inf silently, so it is an error.  bf16
    (DecoderEngine(dtype=torch.bfloat16)): bf16 tensors stay as they are, fp16 / fp32 ones are rounded to bf16.  Quantisation
    scales are always fp16 (the meta image of the W4 / W8 kernels)."""
    if t.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise TypeError(f"{name}: unsupported tensor dtype {t.dtype}")
    if dtype == torch.bfloat16:
        return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)
    if t.dtype == torch.float16:
        return t
    if t.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"{name}: unsupported tensor dtype {t.dtype}")
    o = t.to(torch.float16)
    if not bool(torch.isfinite(o).all()):
        raise ValueError(f"{name}: {t.dtype} values exceed the fp16 range")
    return o


def _linear_fp16(sh: _Shards, prefix: str, dtype: torch.dtype = torch.float16) -> CanonLinear:
    w = _f16(sh.get(prefix + ".weight"), prefix, dtype)    # HF stores [out, in]
    return CanonLinear("fp16", w.shape[1], w.shape[0], w=w.t().contiguous())


def _linear_quant(sh: _Shards, prefix: str, method: str, group_size: int) -> CanonLinear:
    qw, qz, sc = sh.get(prefix + ".qweight"), sh.get(prefix + ".qzeros"), _f16(sh.get(prefix + ".scales"), prefix + ".scales")
    if sh.has(prefix + ".g_idx"):
        gidx = sh.get(prefix + ".g_idx")
        K = gidx.numel()
        if not torch.equal(gidx.to(torch.int64), torch.arange(K) // group_size):
            raise NotImplementedError(f"{prefix}: act-order (non-trivial g_idx) checkpoints are not supported")
    q, z = (quant.unpack_gptq if method == "gptq" else quant.unpack_awq)(qw, qz)
    return CanonLinear("w4", q.shape[0], q.shape[1], q=q, scales=sc, z_eff=z, group_size=group_size)


def load_hf_checkpoint(path: str, quantization: Optional[str] = None, tp: int = 1, rank: int = 0,
                       max_layers: Optional[int] = None, split_embedding: bool = False,
                       dtype: torch.dtype = torch.float16) -> Tuple[ModelConfig, Dict]:
    """Read an HF Qwen2/Llama checkpoint (fp16, GPTQ-4bit or AWQ-4bit) into the canonical weight dict of
    rtp_llm_amd.model (per-rank tensors when tp > 1).  quantization="int8" autoquantises fp16 linears at load time.
    split_embedding (tp > 1): the embedding table comes back as this rank's [vocab, hidden / tp] column slice, the reference's TP
    layout (utils/model_weight.py:1490 sp_neg1) -- pair it with DecoderEngine.set_embedding_split(); default: replicated table
    (no collective in the lookup).  dtype: activation dtype the model will run in (see _f16); bf16 excludes quantization="int8"."""
    if dtype not in (torch.float16, torch.bfloat16) or (dtype == torch.bfloat16 and quantization == "int8"):
        raise NotImplementedError(f"load_hf_checkpoint: dtype {dtype} with quantization {quantization}")
    cfg_json = json.load(open(os.path.join(path, "config.json")))
    mc, qc = config_from_hf(cfg_json, os.path.basename(os.path.normpath(path)))
    if qc["method"] not in ("none", "gptq", "awq") or (qc["method"] != "none" and qc["bits"] != 4):
        raise NotImplementedError(f"quantization_config {qc} is outside the decode path of this build (4-bit GPTQ/AWQ only)")
    if qc["desc_act"]:
        raise NotImplementedError("desc_act=True (act-order) GPTQ checkpoints are not supported (the reference ignores g_idx too)")
    sh = _Shards(path)
    pre = "model." if sh.has("model.embed_tokens.weight") else ""
    L = mc.num_layers if max_layers is None else min(mc.num_layers, max_layers)
    quantised = qc["method"] != "none"

    def lin(name):
        c = _linear_quant(sh, name, qc["method"], qc["group_size"]) if quantised and sh.has(name + ".qweight") else _linear_fp16(sh, name, dtype)
        if c.kind == "fp16" and quantization == "int8":
            q, s = quant.symmetric_quantize_int8(c.w)
            c = CanonLinear("int8", c.K, c.N, q=q, scales=s)
        return c

    def bias(name):
        return _f16(sh.get(name + ".bias"), name + ".bias", dtype) if sh.has(name + ".bias") else None

    layers = []
    has_bias = False
    for i in range(L):
        p = f"{pre}layers.{i}."
        qkv = CanonLinear.cat_cols([lin(p + "self_attn.q_proj"), lin(p + "self_attn.k_proj"), lin(p + "self_attn.v_proj")])
        b = [bias(p + "self_attn.q_proj"), bias(p + "self_attn.k_proj"), bias(p + "self_attn.v_proj")]
        has_bias = has_bias or b[0] is not None
        layer = {
            "qkv": qkv, "o": lin(p + "self_attn.o_proj"),
            "gate_up": CanonLinear.cat_cols([lin(p + "mlp.gate_proj"), lin(p + "mlp.up_proj")]),
            "down": lin(p + "mlp.down_proj"),
            "qkv_bias": torch.cat(b).contiguous() if b[0] is not None else None,
            "input_norm": _f16(sh.get(p + "input_layernorm.weight"), p + "input_layernorm", dtype),
            "post_norm": _f16(sh.get(p + "post_attention_layernorm.weight"), p + "post_attention_layernorm", dtype),
        }
        layers.append(split_layer_tp(layer, mc, tp, rank))
    emb = _f16(sh.get(pre + "embed_tokens.weight"), "embed_tokens", dtype)
    if sh.has("lm_head.weight"):
        head = _f16(sh.get("lm_head.weight"), "lm_head", dtype)
    else:                                                   # tie_word_embeddings (e.g. Qwen2-0.5B)
        head = emb
    V = head.shape[0]                                       # padded-vocab checkpoints: the tensor, not config.json, decides
    if V < mc.vocab or emb.shape[0] < mc.vocab:
        raise ValueError(f"lm_head / embedding rows ({V}, {emb.shape[0]}) < vocab_size {mc.vocab}")
    if tp > 1:                                              # vocab-split lm_head (PyWrappedModel.cc:915-936)
        if V % tp:
            raise ValueError(f"lm_head rows {V} not divisible by tp={tp}")
        head = head[rank * (V // tp):(rank + 1) * (V // tp)]
    mc = ModelConfig(**{**mc.__dict__, "num_layers": L, "qkv_bias": has_bias, "vocab": V})
    if split_embedding and tp > 1:
        from .model import split_embedding_tp
        emb = split_embedding_tp(emb, tp, rank)
    weights = {"layers": layers, "embedding": emb, "final_norm": _f16(sh.get(pre + "norm.weight"), "norm", dtype),
               "lm_head": CanonLinear("fp16", head.shape[1], head.shape[0], w=head.t().contiguous())}
    return (mc.per_rank(tp) if tp > 1 else mc), weights
