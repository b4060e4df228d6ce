"""Model description, canonical weights, TP split and the two ways to run a decode step:

  * ``Qwen2DecoderModel``  — the Python module graph, shaped like the reference's
    Qwen3Model / Qwen3DecoderLayer / CausalAttention / DenseMLP
    (rtp_llm/models_py/model_desc/qwen3.py:57-138, modules/hybrid/causal_attention.py:42-93,
    modules/hybrid/dense_mlp.py:49-106) on top of the plug-in strategies of this package;
  * ``DecoderEngine``      — the C++ step driver (csrc/engine.cpp) that enqueues the same
    step with fused launches and replays it as a hipGraph.

Canonical (pre-packing) weights follow the reference loader's conventions: linear weights are
[K(in), N(out)] (device_impl.py:183-192), QKV merged as [q | k | v] columns (merge_qkv_hf),
gate/up merged as ffn_w13 = [gate | up] (dense_mlp.py:49-70).
"""
import ctypes as C
import math
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _C, ops, quant
from .attention import AttentionConfigs, AttnImplFactory, LayerKVCache, PyAttentionInputs
from .distributed import Group, all_gather, all_reduce
from .kvcache import alloc_layer_cache
from .linear import LinearFactory
from .modules import Embedding, FusedSiluAndMul, RMSNorm
from .quant import PackedWeight


@dataclass
class ModelConfig:
    name: str
    num_layers: int
    hidden: int
    nh: int
    nkv: int
    hd: int
    inter: int
    vocab: int
    rope_theta: float = 1e6
    rms_eps: float = 1e-6
    qkv_bias: bool = True
    max_pos: int = 8192
    # scaled RoPE style, HF's `rope_scaling` dict ({"rope_type": "linear" | "llama3" | "yarn", "factor", ...}); None = base
    # (RopeStyle / RopeConfig, rtp_llm/cpp/model_utils/RopeConfig.h:7-40; mapping from config.json: models/llama.py:87-117)
    rope_scaling: Optional[dict] = None

    def per_rank(self, tp: int) -> "ModelConfig":
        """Per-rank attention/FFN dims (model_config.getAttentionConfigs(tp), qwen3.py:33;
        K/V heads split by tp when divisible, utils/model_weight.py:447-466)."""
        if tp == 1:
            return self
        if self.nh % tp or self.vocab % tp or (self.nkv % tp and tp % self.nkv):
            raise ValueError(f"{self.name}: nh={self.nh} / vocab={self.vocab} must be divisible by tp={tp}, and nkv={self.nkv} "
                             f"must divide or be divisible by it")
        return replace(self, nh=self.nh // tp, nkv=self.kv_heads_per_rank(tp), inter=self.padded_inter(tp) // tp,
                       vocab=self.vocab // tp)

    def kv_heads_per_rank(self, tp: int) -> int:
        """K/V heads are split by tp while there are enough of them and replicated beyond that (get_sp_tensor splits
        K/V by gcd(nkv, tp), utils/model_weight.py:447-466): Qwen2-7B (nkv = 4) at tp = 8 keeps one kv head per rank,
        shared by the two ranks that own its query heads."""
        return self.nkv // tp if self.nkv >= tp else 1

    def kv_head_of_rank(self, tp: int, rank: int) -> int:
        """First kv head of `rank` (the only one when heads are replicated)."""
        return rank * self.nkv // tp

    def padded_inter(self, tp: int) -> int:
        """FFN width padded so that every rank gets a whole number of 128-row quantisation groups / K chunks
        (the reference's align_size = tp * group_size, utils/model_weight.py:234-250): Qwen2-72B's 29568 = 231 * 128
        becomes 29696 at tp = 8.  Padded columns / rows carry zero weights."""
        a = tp * 128
        return (self.inter + a - 1) // a * a


# dims from the HF config.json values quoted in SURVEY section 8 (models/qwen_v2.py:338-399 reads them)
QWEN2_7B = ModelConfig("qwen2-7b", 28, 3584, 28, 4, 128, 18944, 152064)
QWEN2_0_5B = ModelConfig("qwen2-0.5b", 24, 896, 14, 2, 64, 4864, 151936)
LLAMA3_70B = ModelConfig("llama3-70b", 80, 8192, 64, 8, 128, 28672, 128256, rope_theta=5e5, qkv_bias=False)
QWEN2_72B = ModelConfig("qwen2-72b", 80, 8192, 64, 8, 128, 29568, 152064)
MODELS = {m.name: m for m in (QWEN2_7B, QWEN2_0_5B, LLAMA3_70B, QWEN2_72B)}


# --------------------------------------------------------------------------- canonical weights
@dataclass
class CanonLinear:
    """One linear layer before packing.  kind: 'fp16' | 'int8' | 'w4'."""
    kind: str
    K: int
    N: int
    w: Optional[torch.Tensor] = None        # fp16 [K,N]            (fp16)
    q: Optional[torch.Tensor] = None        # int8 [K,N] / uint8 codes [K,N]
    scales: Optional[torch.Tensor] = None   # [N] (int8) / fp16 [K/g, N] (w4)
    z_eff: Optional[torch.Tensor] = None    # [K/g, N] effective zero codes (w4)
    group_size: int = 0

    def pack(self, gate_up: bool = False, dtype: torch.dtype = torch.float16) -> PackedWeight:
        """dtype: activation dtype the packed weight will run with -- only a 16-bit weight image depends on it (its elements are
        converted); W4 / W8 images serve fp16 and bf16 activations alike."""
        il = (lambda t: quant.interleave_gate_up(t, -1)) if gate_up else (lambda t: t)
        if self.kind == "fp16":
            return quant.pack_fp16(il(self.w).to(dtype))
        if self.kind == "int8":
            return quant.pack_int8_per_channel(il(self.q), il(self.scales))
        return quant.pack_groupwise_w4(il(self.q), il(self.z_eff), il(self.scales), self.group_size)

    def cols(self, lo: int, hi: int) -> "CanonLinear":
        """Column (output) slice — column-parallel split (sp_head / ffn_sp_neg1, utils/model_weight.py:265-277,472-509)."""
        s = lambda t: None if t is None else t[..., lo:hi].contiguous()
        return CanonLinear(self.kind, self.K, hi - lo, s(self.w), s(self.q), s(self.scales), s(self.z_eff), self.group_size)

    def rows(self, lo: int, hi: int) -> "CanonLinear":
        """Row (input) slice — row-parallel split (sp_0 / ffn_sp_0, :234-250); group aligned."""
        s = lambda t: None if t is None else t[lo:hi].contiguous()
        if self.kind == "w4":
            g = self.group_size
            assert lo % g == 0 and hi % g == 0, "row split must be aligned to the quantisation group (align_size = tp*g)"
            return CanonLinear(self.kind, hi - lo, self.N, None, s(self.q), self.scales[lo // g: hi // g].contiguous(),
                               self.z_eff[lo // g: hi // g].contiguous(), g)
        return CanonLinear(self.kind, hi - lo, self.N, s(self.w), s(self.q), self.scales, None, 0)

    def pad_cols(self, n: int) -> "CanonLinear":
        """Append n output columns of zero weight (code 0, zero 0, scale 0)."""
        if n == 0:
            return self
        z = lambda t: None if t is None else torch.cat([t, torch.zeros(*t.shape[:-1], n, dtype=t.dtype, device=t.device)], dim=-1)
        return CanonLinear(self.kind, self.K, self.N + n, z(self.w), z(self.q), z(self.scales), z(self.z_eff), self.group_size)

    def pad_rows(self, n: int) -> "CanonLinear":
        """Append n input rows of zero weight; for w4 n must be a multiple of the group size (new groups: scale 0)."""
        if n == 0:
            return self
        z = lambda t, k: None if t is None else torch.cat([t, torch.zeros(k, *t.shape[1:], dtype=t.dtype, device=t.device)], dim=0)
        if self.kind == "w4":
            assert n % self.group_size == 0
            return CanonLinear(self.kind, self.K + n, self.N, None, z(self.q, n), z(self.scales, n // self.group_size),
                               z(self.z_eff, n // self.group_size), self.group_size)
        return CanonLinear(self.kind, self.K + n, self.N, z(self.w, n), z(self.q, n), self.scales, None, 0)

    @staticmethod
    def cat_cols(parts: List["CanonLinear"]) -> "CanonLinear":
        c = lambda name: None if getattr(parts[0], name) is None else torch.cat([getattr(p, name) for p in parts], dim=-1)
        p0 = parts[0]
        return CanonLinear(p0.kind, p0.K, sum(p.N for p in parts), c("w"), c("q"), c("scales"), c("z_eff"), p0.group_size)


def synth_linear(K: int, N: int, kind: str, device, gen: torch.Generator, group_size: int = 128, method: str = "gptq",
                 zeros: str = "uniform") -> CanonLinear:
    """Synthetic weights of SURVEY 8d: fp16 W ~ xavier_uniform; GPTQ/AWQ: q ~ U{0..15}, z ~ U{0..15},
    scale ~ U(0.5,1.5) * (2*amax/15); INT8: autoquant of the fp16 weights via a1.
    zeros="centered" draws the EFFECTIVE zero from {7, 8} instead, so that E[q - z_eff] = 0 (what real GPTQ/AWQ
    checkpoints of zero-mean weights look like).  With z ~ U{0..15} (and GPTQ's +1) the weights have mean -scale: the
    all-ones direction then has gain K * |mean W| (41 for the 7B down-proj), the residual stream of a 3584-wide model
    reaches +-2000 after two layers and the q.k logits turn the softmax into a chaotic arg-max -- fine for timing,
    meaningless for end-to-end logits parity."""
    a = math.sqrt(6.0 / (K + N))
    if kind == "fp16" or kind == "int8":
        w = ((torch.rand(K, N, device=device, generator=gen) * 2 - 1) * a).half()
        if kind == "fp16":
            return CanonLinear("fp16", K, N, w=w)
        q, s = quant.symmetric_quantize_int8(w)
        return CanonLinear("int8", K, N, q=q, scales=s)
    G = K // group_size
    q = torch.randint(0, 16, (K, N), device=device, generator=gen, dtype=torch.uint8)
    flag = 1 if method == "gptq" else 0
    zlo, zhi = (7 - flag, 9 - flag) if zeros == "centered" else (0, 16)
    z = torch.randint(zlo, zhi, (G, N), device=device, generator=gen, dtype=torch.uint8)
    z_eff = (z.to(torch.int16) + (1 if method == "gptq" else 0)).to(torch.uint8)
    scales = ((torch.rand(G, N, device=device, generator=gen) + 0.5) * (2 * a / 15)).half()
    return CanonLinear("w4", K, N, q=q, scales=scales, z_eff=z_eff, group_size=group_size)


def synth_layer(cfg: ModelConfig, kind: str, device, gen, group_size=128, method="gptq", zeros="uniform") -> Dict:
    H, qkv_n = cfg.hidden, (cfg.nh + 2 * cfg.nkv) * cfg.hd
    lin = lambda K, N: synth_linear(K, N, kind, device, gen, group_size, method, zeros)
    return {
        "qkv": lin(H, qkv_n), "o": lin(cfg.nh * cfg.hd, H), "gate_up": lin(H, 2 * cfg.inter), "down": lin(cfg.inter, H),
        "qkv_bias": ((torch.rand(qkv_n, device=device, generator=gen) - 0.5) * 0.2).half() if cfg.qkv_bias else None,
        "input_norm": (1.0 + 0.1 * torch.randn(H, device=device, generator=gen)).half(),
        "post_norm": (1.0 + 0.1 * torch.randn(H, device=device, generator=gen)).half(),
    }


def synth_model(cfg: ModelConfig, kind: str, device, seed: int = 0, group_size=128, method="gptq", zeros="uniform") -> Dict:
    gen = torch.Generator(device=device).manual_seed(seed)
    return {
        "layers": [synth_layer(cfg, kind, device, gen, group_size, method, zeros) for _ in range(cfg.num_layers)],
        "embedding": (torch.randn(cfg.vocab, cfg.hidden, device=device, generator=gen) * 0.5).half(),
        "final_norm": (1.0 + 0.1 * torch.randn(cfg.hidden, device=device, generator=gen)).half(),
        "lm_head": synth_linear(cfg.hidden, cfg.vocab, "fp16", device, gen),
    }


def weights_to(w: Dict, device) -> Dict:
    """Move a canonical weight dict (synth_model layout) to `device`."""
    def mv(v):
        if torch.is_tensor(v):
            return v.to(device)
        if isinstance(v, CanonLinear):
            return CanonLinear(v.kind, v.K, v.N, *(None if t is None else t.to(device) for t in (v.w, v.q, v.scales, v.z_eff)),
                               v.group_size)
        if isinstance(v, dict):
            return {k: mv(x) for k, x in v.items()}
        if isinstance(v, list):
            return [mv(x) for x in v]
        return v
    return mv(w)


def split_embedding_tp(embedding: torch.Tensor, tp: int, rank: int) -> torch.Tensor:
    """This rank's hidden-dimension slice [vocab, hidden / tp] of the embedding table -- the TP layout of the reference's
    embedding weight (utils/model_weight.py:257-263 sp_neg1, :1490; consumed by modules/base/common/embedding.py:22-59)."""
    H = embedding.shape[1]
    if H % (8 * tp):
        raise ValueError(f"hidden {H} does not split into {tp} slices of whole 16-byte vectors")
    n = H // tp
    return embedding[:, rank * n:(rank + 1) * n].contiguous()


def split_layer_tp(layer: Dict, cfg: ModelConfig, tp: int, rank: int) -> Dict:
    """Megatron TP split of one layer (table utils/model_weight.py:1517-1563): column-parallel QKV
    (q heads / tp, k and v heads / tp) and gate/up, row-parallel O and down; norms replicated;
    QKV bias split like the QKV columns."""
    if tp == 1:
        return layer
    hd, nh, nkv, I0 = cfg.hd, cfg.nh, cfg.nkv, cfg.inter
    I = cfg.padded_inter(tp)                      # align_size padding: zero columns of gate / up, zero rows of down
    nh_r, nkv_r, I_r = nh // tp, cfg.kv_heads_per_rank(tp), I // tp
    qkv = layer["qkv"]
    kv0 = cfg.kv_head_of_rank(tp, rank)           # tp > nkv: the kv head is replicated over tp / nkv ranks
    q_lo, k_lo, v_lo = rank * nh_r * hd, (nh + kv0) * hd, (nh + nkv + kv0) * hd
    parts = [qkv.cols(q_lo, q_lo + nh_r * hd), qkv.cols(k_lo, k_lo + nkv_r * hd), qkv.cols(v_lo, v_lo + nkv_r * hd)]
    gu, down = layer["gate_up"], layer["down"]
    if I != I0:
        gu = CanonLinear.cat_cols([gu.cols(0, I0).pad_cols(I - I0), gu.cols(I0, 2 * I0).pad_cols(I - I0)])
        down = down.pad_rows(I - I0)
    out = {
        "qkv": CanonLinear.cat_cols(parts),
        "o": layer["o"].rows(rank * nh_r * hd, (rank + 1) * nh_r * hd),
        "gate_up": CanonLinear.cat_cols([gu.cols(rank * I_r, (rank + 1) * I_r), gu.cols(I + rank * I_r, I + (rank + 1) * I_r)]),
        "down": down.rows(rank * I_r, (rank + 1) * I_r),
        "input_norm": layer["input_norm"], "post_norm": layer["post_norm"], "qkv_bias": None,
    }
    if layer["qkv_bias"] is not None:
        b = layer["qkv_bias"]
        out["qkv_bias"] = torch.cat([b[q_lo:q_lo + nh_r * hd], b[k_lo:k_lo + nkv_r * hd], b[v_lo:v_lo + nkv_r * hd]]).contiguous()
    return out


def rope_frequencies(hd: int, theta: float, scaling: Optional[dict]):
    """(angular step per channel pair [hd / 2] fp32, cos / sin multiplier) of the configured RoPE style.  The kernels only see
    the position-indexed table, so a style is whatever can be folded into it:
      * base, linear (positions / factor)                            LinearScaleRope, rotary_position_embedding.h:355-364
      * llama3: long wavelengths / factor, short ones kept, a linear blend between the two bands      Llama3Rope, :418-442
      * yarn: blend of interpolated (1 / factor) and original steps along a ramp over the channel index, whose ends come from
        beta_fast / beta_slow rotations over the original context; table scaled by 0.1 ln(factor) + 1      YarnRope, :366-416
    Dynamic-NTK styles (:889-902) change the base with the cached length -- for decode that IS the position: rope_table builds their
    rows one base per position (dynamic_ntk_base); they have no single step per channel and are not served here."""
    step = 1.0 / torch.pow(torch.tensor(float(theta)), torch.arange(0, hd, 2).float() / hd)
    kind = (scaling or {}).get("rope_type", (scaling or {}).get("type"))
    if kind in (None, "default"):
        return step, 1.0
    factor = float(scaling.get("factor", 1.0))
    if kind == "linear":
        return step / factor, 1.0
    if kind == "llama3":
        ctx = float(scaling["original_max_position_embeddings"])
        lo, hi = float(scaling["low_freq_factor"]), float(scaling["high_freq_factor"])
        wavelen = 2 * math.pi / step
        blend = (ctx / wavelen - lo) / (hi - lo)
        banded = (1 - blend) * step / factor + blend * step
        return torch.where(wavelen < ctx / hi, step, torch.where(wavelen > ctx / lo, step / factor, banded)), 1.0
    if kind == "yarn":
        ctx = int(scaling["original_max_position_embeddings"])
        fast, slow = int(scaling.get("beta_fast", 32)), int(scaling.get("beta_slow", 1))
        chan = lambda rotations: hd * math.log(ctx / (rotations * 2 * math.pi)) / (2 * math.log(int(theta)))
        first, last = float(max(math.floor(chan(fast)), 0)), float(min(math.ceil(chan(slow)), hd - 1))
        if first == last:
            last += 0.001
        keep = (1 - torch.clamp((torch.arange(hd // 2).float() - first) / (last - first), 0, 1)) * float(scaling.get("extrapolation_factor", 1.0))
        gain = 0.1 * math.log(factor) + 1.0 if factor > 1 else 1.0
        return (step / factor) * (1 - keep) + step * keep, gain
    raise NotImplementedError(f"rope_scaling type {kind!r}: no single step per channel (base / linear / llama3 / yarn have one; the dynamic-NTK "
                              "styles go through dynamic_ntk_base)")


DYNAMIC_NTK = ("dynamic", "qwen_dynamic")


def dynamic_ntk_base(hd: int, theta: float, positions: torch.Tensor, scaling: dict) -> torch.Tensor:
    """fp32 base of the rotation at each decode position under RopeStyle::DynamicNTK ("dynamic") / QwenDynamicNTK ("qwen_dynamic"),
    rotary_position_embedding.h:889-902 + :925-951.  The decode writer hands apply_rope the cached length as seq_len
    (fused_rope_kvcache_kernel.cu:1341-1392), i.e. the new token's position: past the original context the base grows with it."""
    pos = positions.float()
    ctx = float(int(scaling["original_max_position_embeddings"]))
    power = torch.tensor(hd / (hd - 2.0), dtype=torch.float32)
    theta_t = torch.full_like(pos, float(theta))
    kind = scaling.get("rope_type", scaling.get("type"))
    if kind == "dynamic":
        f = torch.tensor(float(scaling.get("factor", 1.0)), dtype=torch.float32)
        stretched = theta_t * torch.pow(f * pos / ctx - (f - 1.0), power)
    else:
        octave = torch.ceil(torch.log(torch.clamp(pos, min=1.0) / ctx) / math.log(2.0) + 1.0)
        stretched = theta_t * torch.pow(torch.clamp(torch.exp2(octave) - 1.0, min=1.0), power)
    return torch.where(pos > ctx, stretched, theta_t)


def prefill_rope_table_dynamic_ntk(cfg: ModelConfig, seq_len: int, device) -> torch.Tensor:
    """{cos, sin} rows [max_pos][hd / 2][2] of a PREFILL batch under a dynamic-NTK style: one base for every position, that of `seq_len` = the longest
    prompt of the batch (context_rope -> apply_rope(..., seq_len), rotary_position_embedding.h:925-951,1000-1025)."""
    pos = torch.arange(cfg.max_pos)
    base = dynamic_ntk_base(cfg.hd, cfg.rope_theta, torch.tensor([seq_len]), cfg.rope_scaling)[0]
    chan = torch.arange(0, cfg.hd, 2).float() / cfg.hd
    angle = pos.float()[:, None] / torch.pow(base, chan[None, :])
    return torch.stack((angle.cos(), angle.sin()), dim=-1).contiguous().to(device)


def rope_table(cfg: ModelConfig, device) -> torch.Tensor:
    """fp32 {cos,sin} table [max_pos][hd/2][2] (genBaseCache / genYarnCache, cpp/model_utils/RopeCache.cc:16-81; the styles the
    reference computes in-kernel on ROCm are tabulated the same way: rope_frequencies); built on the host in fp32 so every
    rank holds identical bits."""
    kind = (cfg.rope_scaling or {}).get("rope_type", (cfg.rope_scaling or {}).get("type"))
    if kind in DYNAMIC_NTK:   # one base per position (decode: the cached length is the position)
        pos = torch.arange(cfg.max_pos)
        chan = torch.arange(0, cfg.hd, 2).float() / cfg.hd
        angle = pos.float()[:, None] / torch.pow(dynamic_ntk_base(cfg.hd, cfg.rope_theta, pos, cfg.rope_scaling)[:, None], chan[None, :])
        return torch.stack((angle.cos(), angle.sin()), dim=-1).contiguous().to(device)
    step, gain = rope_frequencies(cfg.hd, cfg.rope_theta, cfg.rope_scaling)
    freqs = torch.outer(torch.arange(cfg.max_pos).float(), step)
    if gain != 1.0:
        return torch.stack((freqs.cos() * gain, freqs.sin() * gain), dim=-1).contiguous().to(device)
    return torch.stack((freqs.cos(), freqs.sin()), dim=-1).contiguous().to(device)


# --------------------------------------------------------------------------- Python module graph
class CausalAttention(nn.Module):
    """causal_attention.py:42-93: qkv_proj -> fmha_impl.forward -> o_proj -> all_reduce(TP)."""

    def __init__(self, layer: Dict):
        super().__init__()
        mk = lambda c, bias=None: LinearFactory.create_linear(c.w if c.kind == "fp16" else c.q, bias, c.scales, None, c.z_eff)
        self.qkv_proj, self.o_proj = mk(layer["qkv"], layer["qkv_bias"]), mk(layer["o"])

    def forward(self, x, fmha_impl, kv_cache: LayerKVCache, layer_idx: int):
        out = self.o_proj(fmha_impl.forward(self.qkv_proj(x), kv_cache, layer_idx))
        return all_reduce(out, Group.TP)


class DenseMLP(nn.Module):
    """dense_mlp.py:95-106: up_proj (merged gate_up) -> FusedSiluAndMul -> down_proj -> all_reduce(TP)."""

    def __init__(self, layer: Dict):
        super().__init__()
        mk = lambda c: LinearFactory.create_linear(c.w if c.kind == "fp16" else c.q, None, c.scales, None, c.z_eff)
        self.gate_up_proj, self.down_proj, self.act = mk(layer["gate_up"]), mk(layer["down"]), FusedSiluAndMul()

    def forward(self, x):
        return all_reduce(self.down_proj(self.act(self.gate_up_proj(x))), Group.TP)


class Qwen2DecoderModel(nn.Module):
    """qwen3.py:82-138 for the Qwen2/Llama family; forward(input_ids, attn_inputs, kv_caches) -> hidden."""

    def __init__(self, cfg: ModelConfig, weights: Dict, page: int, max_seq_len: int):
        super().__init__()
        self.cfg, self.page = cfg, page
        dev = weights["embedding"].device
        self.embed = Embedding(weights["embedding"])
        self.layers = nn.ModuleList()
        for L in weights["layers"]:
            m = nn.Module()
            m.input_layernorm, m.post_attention_layernorm = RMSNorm(L["input_norm"], cfg.rms_eps), RMSNorm(L["post_norm"], cfg.rms_eps)
            m.self_attn, m.mlp = CausalAttention(L), DenseMLP(L)
            self.layers.append(m)
        self.norm = RMSNorm(weights["final_norm"], cfg.rms_eps)
        self.lm_head = LinearFactory.create_linear(weights["lm_head"].w)
        self.attn_cfg = AttentionConfigs(cfg.nh, cfg.nkv, cfg.hd, cfg.hd, cfg.rope_theta, max_seq_len, 1.0, page)
        self.cos_sin = rope_table(cfg, dev)

    def prepare_fmha_impl(self, attn_inputs: PyAttentionInputs):   # module_base.py:77-109
        return AttnImplFactory.get_fmha_impl(self.attn_cfg, attn_inputs, None, cos_sin=self.cos_sin)

    def forward(self, input_ids: torch.Tensor, fmha_impl, kv_caches: List[LayerKVCache]) -> torch.Tensor:
        h = self.embed(input_ids)
        for i, m in enumerate(self.layers):           # Qwen3DecoderLayer.forward, qwen3.py:57-79
            r = h
            h = m.self_attn(m.input_layernorm(h), fmha_impl, kv_caches[i], i)
            h = r + h
            r = h
            h = m.mlp(m.post_attention_layernorm(h))
            h = r + h
        return self.norm(h)

    def logits(self, hidden: torch.Tensor) -> torch.Tensor:   # PyWrappedModel.cc:1039-1047
        return all_gather(ops.linear(hidden, self.lm_head.packed, None, epilogue=_C.EPI_OUT_F32), Group.TP)


# --------------------------------------------------------------------------- C++ step driver
class DecoderEngine:
    """Owns the device buffers and drives csrc/engine.cpp.  Greedy decode only (top_k = 1)."""

    def __init__(self, cfg: ModelConfig, weights: Dict, *, kv_int8: bool, page: int, num_blocks: int, max_batch: int,
                 max_seq_len: int, device, tp_size: int = 1, vocab_full: Optional[int] = None, dtype: torch.dtype = torch.float16):
        """dtype: activation dtype of the step (torch.float16 or torch.bfloat16; the reference runs either).  bf16: every 16-bit
        tensor of the model (embedding, norm weights, biases, a 16-bit lm_head / linear) is converted once here, the KV cache is
        bf16 (or INT8), W4 / W8 weights are shared as they are."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"DecoderEngine: dtype {dtype}")
        bf = dtype == torch.bfloat16
        self.dtype = dtype
        def cast(t):
            # every 16-bit tensor of the model takes the engine's activation dtype: bf16 checkpoint tensors handed to an fp16 engine (or
            # the other way round) are converted, and an fp16 overflow raises instead of feeding inf / garbage bits to the kernels
            if t is None or t.dtype == dtype:
                return t
            if t.dtype not in (torch.float16, torch.bfloat16):
                raise ValueError(f"DecoderEngine: a {t.dtype} tensor where a 16-bit one of the model is expected")
            out = t.to(dtype)
            if not torch.isfinite(out.float()).all():
                raise ValueError(f"DecoderEngine: converting a {t.dtype} tensor to {dtype} overflows; load the checkpoint in the engine's dtype")
            return out
        self.cfg, self.device, self.tp_size = cfg, device, tp_size
        self.page, self.num_blocks, self.max_batch, self.max_seq_len = page, num_blocks, max_batch, max_seq_len
        self.max_blocks_per_seq = (max_seq_len + page - 1) // page
        self.lib = _C.lib()
        self._keep = []   # keep packed tensors alive
        self.packed_bytes = 0
        lw = (_C.LayerWeights * cfg.num_layers)()
        self.kv, self.kv_scale = [], []
        for i, L in enumerate(weights["layers"]):
            p = {k: L[k].pack(gate_up=(k == "gate_up"), dtype=dtype) for k in ("qkv", "o", "gate_up", "down")}
            aux = {k: cast(L[k]) for k in ("qkv_bias", "input_norm", "post_norm")}
            self._keep.append((p, L, aux))
            self.packed_bytes += sum(v.nbytes for v in p.values())
            kvb, kvs = alloc_layer_cache(num_blocks, cfg.nkv, page, cfg.hd, kv_int8, device, dtype=dtype)
            self.kv.append(kvb); self.kv_scale.append(kvs)
            lw[i].qkv, lw[i].o = ops.weight_struct(p["qkv"], dtype), ops.weight_struct(p["o"], dtype)
            lw[i].gate_up, lw[i].down = ops.weight_struct(p["gate_up"], dtype), ops.weight_struct(p["down"], dtype)
            lw[i].qkv_bias = 0 if aux["qkv_bias"] is None else aux["qkv_bias"].data_ptr()
            lw[i].input_norm, lw[i].post_norm = aux["input_norm"].data_ptr(), aux["post_norm"].data_ptr()
            lw[i].kv_base = kvb.data_ptr()
            lw[i].kv_scale_base = 0 if kvs is None else kvs.data_ptr()
        self.lm_head = weights["lm_head"].pack(dtype=dtype) if isinstance(weights["lm_head"], CanonLinear) else weights["lm_head"]
        self.packed_bytes_lm_head = self.lm_head.nbytes
        self.embedding, self.final_norm = cast(weights["embedding"]), cast(weights["final_norm"])
        self.cos_sin = rope_table(cfg, device)
        mc = _C.ModelConfig(cfg.num_layers, cfg.hidden, cfg.nh, cfg.nkv, cfg.hd, cfg.inter, cfg.vocab, cfg.hd, cfg.max_pos,
                            cfg.rms_eps, _C.KV_INT8 if kv_int8 else (_C.KV_BF16 if bf else _C.KV_FP16), page, num_blocks, max_batch,
                            self.max_blocks_per_seq, max_seq_len, tp_size, _C.ACT_BF16 if bf else _C.ACT_F16)
        mw = _C.ModelWeights(self.embedding.data_ptr(), vocab_full or self.embedding.shape[0], self.final_norm.data_ptr(),
                             ops.weight_struct(self.lm_head, dtype), self.cos_sin.data_ptr())
        i32 = dict(dtype=torch.int32, device=device)
        self.token_ids = torch.zeros(max_batch, **i32)
        self.positions = torch.zeros(max_batch, **i32)
        self.block_table = torch.zeros(max_batch, self.max_blocks_per_seq, **i32)
        self.logits = torch.zeros(max_batch, cfg.vocab, dtype=torch.float32, device=device)
        self.hidden = torch.zeros(max_batch, cfg.hidden, dtype=dtype, device=device)
        self.ar_buf = torch.zeros(max_batch, cfg.hidden, dtype=dtype, device=device)
        ws_bytes = self.lib.mi355_decoder_workspace_bytes(C.byref(mc))
        self.workspace = torch.zeros(ws_bytes, dtype=torch.uint8, device=device)
        sb = _C.StepBuffers(self.token_ids.data_ptr(), self.positions.data_ptr(), self.block_table.data_ptr(),
                            self.logits.data_ptr(), self.hidden.data_ptr(), self.ar_buf.data_ptr(), self.workspace.data_ptr(), ws_bytes)
        self._structs = (mc, mw, sb, lw)
        self.handle = self.lib.mi355_decoder_create(C.byref(mc), lw, C.byref(mw), C.byref(sb))
        if not self.handle:
            raise _C.Mi355Error("decoder_create failed: " + self.lib.mi355_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.mi355_decoder_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- inputs
    def set_inputs(self, token_ids, positions, block_table):
        B = len(token_ids)
        self.token_ids[:B].copy_(torch.as_tensor(token_ids, dtype=torch.int32))
        self.positions[:B].copy_(torch.as_tensor(positions, dtype=torch.int32))
        bt = torch.as_tensor(block_table, dtype=torch.int32)
        self.block_table[:bt.shape[0], : bt.shape[1]].copy_(bt)   # one row per token (decode) or per sequence (q_len > 1)

    def _st(self) -> int:
        return torch.cuda.current_stream().cuda_stream

    def capacity(self, block_table=None) -> int:
        """Tokens one sequence may hold: bounded by the engine's max_seq_len, the rotation table and the block table."""
        cap = min(self.max_seq_len, self.cfg.max_pos, self.max_blocks_per_seq * self.page)
        if block_table is not None:
            cap = min(cap, int(torch.as_tensor(block_table).shape[1]) * self.page)
        return cap

    def check_room(self, ctx_lens, new_tokens: int, block_table=None, what: str = "decode"):
        """Raise before anything is enqueued if some sequence would outgrow its cache (graph replay advances positions on
        the device, so nothing else would notice; the KV writer drops out-of-range tokens and counts them, see
        oob_count())."""
        cap = self.capacity(block_table)
        worst = max(int(c) for c in ctx_lens) + int(new_tokens)
        if worst > cap:
            raise _C.Mi355Error(f"{what}: a sequence would reach {worst} tokens, capacity is {cap} "
                                f"(max_seq_len {self.max_seq_len}, max_pos {self.cfg.max_pos}, block table {self.max_blocks_per_seq} x {self.page})")

    def oob_count(self) -> int:
        """Tokens the KV writer refused since creation (stale position / block id).  Synchronises.  Under tensor parallelism it
        also raises when a peer never arrived at an all-reduce within the kernel's spin bound (the step's outputs are then
        garbage, not an error code: checked wherever the out-of-range count is)."""
        ar = getattr(self, "_ar", None)
        if ar is not None:
            st = ar.status()
            if st:
                raise _C.Mi355Error(f"tensor-parallel all-reduce: a peer did not arrive within the spin bound (status {st}); results are invalid")
        n = self.lib.mi355_decoder_oob_count(self.handle, self._st())
        if n < 0:
            _C.check(int(n), "decoder_oob_count")
        return int(n)

    # ---- tp = 1
    def step(self, B: int):
        _C.check(self.lib.mi355_decoder_step(self.handle, B, self._st()), "decoder_step")

    def forward(self, B: int, q_len: int = 1):
        """The decode step without sampling: logits of the B rows stay in `self.logits`, token_ids / positions are not
        advanced (speculative verify, scoring).  q_len > 1: B = nseq * q_len rows, q_len consecutive rows per sequence,
        block_table one row per sequence; the rows of a sequence share one pass over its KV (causal inside the kernel)."""
        st, h, lib = self._st(), self.handle, self.lib
        _C.check(lib.mi355_decoder_begin_rows(h, B // q_len, q_len, st), "decoder_begin")
        for l in range(self.cfg.num_layers):
            _C.check(lib.mi355_decoder_layer_attn(h, l, st), "decoder_layer_attn")
            _C.check(lib.mi355_decoder_layer_mlp(h, l, st), "decoder_layer_mlp")
        _C.check(lib.mi355_decoder_finish(h, 0, st), "decoder_finish")

    # ---- whole-request helpers
    def ingest(self, prompts: List[List[int]], block_table) -> List[int]:
        """Write the K/V of every prompt token except the last into the paged cache and return the context lengths.
        A prompt token is a decode row with its own position and its sequence's block table, so up to `max_batch`
        tokens (of any mix of sequences) go through one step and causality holds by construction: row (b, p) attends to
        positions <= p of sequence b, all written by this or an earlier step (rope_kv_write precedes attention).
        Chunked prefill at decode-kernel efficiency (kept for tests and tiny prompts); `prefill` below is the real path."""
        bt = torch.as_tensor(block_table, dtype=torch.int32)
        self.check_room([len(pr) for pr in prompts], 0, bt, "ingest")
        rows = [(b, tok, pos) for b, pr in enumerate(prompts) for pos, tok in enumerate(pr[:-1])]
        # a row may only run once all earlier positions of its sequence are in the cache or in the same step: order by
        # position, so every step holds a prefix-closed set
        rows.sort(key=lambda r: r[2])
        for i in range(0, len(rows), self.max_batch):
            chunk = rows[i:i + self.max_batch]
            self.set_inputs([r[1] for r in chunk], [r[2] for r in chunk], bt[[r[0] for r in chunk]])
            self.forward(len(chunk))
        return [len(pr) - 1 for pr in prompts]

    def prefill(self, prompts: List[List[int]], block_table, chunk: int = 512, start: Optional[List[int]] = None) -> torch.Tensor:
        """Real prefill (csrc/engine.cpp mi355_decoder_prefill): the prompts go through the large-M GEMMs and the causal
        multi-row attention in chunks of `chunk` tokens per sequence (ragged prompts are padded with position -1 rows).
        `start[b]` tokens of sequence b are already cached.  Returns the fp32 logits [nseq, vocab] of every sequence's LAST
        prompt token (its K/V is in the cache afterwards: decoding continues with the sampled token at position len)."""
        nseq = len(prompts)
        start = [0] * nseq if start is None else list(start)
        rs = self.cfg.rope_scaling or {}
        ntk_table = None
        if rs.get("rope_type", rs.get("type")) in DYNAMIC_NTK and max(len(p) for p in prompts) > int(rs["original_max_position_embeddings"]):
            # context_rope rotates EVERY token of the prefill batch with ONE base, that of the batch's longest prompt (rotary_position_embedding.h:1000-1025,
            # fused_rope_kvcache_kernel.cu:219-260): a table of its own for these chunks (round 6; the decode steps that follow use the position-indexed one)
            if any(start):
                raise NotImplementedError("dynamic-NTK RoPE: a prompt past the original context on top of cached tokens (prefix reuse) is not served")
            ntk_table = prefill_rope_table_dynamic_ntk(self.cfg, max(len(p) for p in prompts), self.device)
            _C.check(self.lib.mi355_decoder_set_prefill_rope_table(self.handle, ntk_table.data_ptr()), "decoder_set_prefill_rope_table")
        try:
            return self._prefill_chunks(prompts, block_table, chunk, start)
        finally:
            if ntk_table is not None:
                torch.cuda.synchronize()      # the chunks read the table: it may only go once they are done
                _C.check(self.lib.mi355_decoder_set_prefill_rope_table(self.handle, None), "decoder_set_prefill_rope_table")

    def _prefill_chunks(self, prompts, block_table, chunk, start):
        nseq = len(prompts)
        bt = torch.as_tensor(block_table, dtype=torch.int32)
        self.check_room([s + len(p) for s, p in zip(start, prompts)], 0, bt, "prefill")
        dev, st = self.device, self._st()
        btd = torch.zeros(nseq, self.max_blocks_per_seq, dtype=torch.int32, device=dev)
        btd[:, : bt.shape[1]].copy_(bt)
        longest = max(len(p) for p in prompts)
        logits = torch.zeros(nseq, self.cfg.vocab, dtype=torch.float32, device=dev)
        done = torch.zeros(nseq, dtype=torch.bool)
        for c0 in range(0, longest, chunk):
            q_len = min(chunk, longest - c0)
            toks = torch.zeros(nseq, q_len, dtype=torch.int32)
            pos = torch.full((nseq, q_len), -1, dtype=torch.int32)
            last = torch.zeros(nseq, dtype=torch.int32)
            need_logits = torch.zeros(nseq, dtype=torch.bool)
            for b, p in enumerate(prompts):
                n = max(0, min(q_len, len(p) - c0))
                if n:
                    toks[b, :n] = torch.tensor(p[c0:c0 + n], dtype=torch.int32)
                    pos[b, :n] = torch.arange(start[b] + c0, start[b] + c0 + n, dtype=torch.int32)
                    if c0 + n == len(p):
                        last[b], need_logits[b] = b * q_len + n - 1, True
            T = nseq * q_len
            ws_bytes = self.lib.mi355_decoder_prefill_workspace_bytes(self.handle, T, nseq)
            if getattr(self, "_prefill_ws", None) is None or self._prefill_ws.numel() < ws_bytes:
                self._prefill_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            tmp_logits = torch.empty(nseq, self.cfg.vocab, dtype=torch.float32, device=dev) if bool(need_logits.any()) else None
            td, pd, ld = toks.reshape(-1).to(dev), pos.reshape(-1).to(dev), last.to(dev)
            _C.check(self.lib.mi355_decoder_prefill(self.handle, td.data_ptr(), pd.data_ptr(), btd.data_ptr(), nseq, q_len,
                                                    ld.data_ptr() if tmp_logits is not None else None,
                                                    tmp_logits.data_ptr() if tmp_logits is not None else None,
                                                    self._prefill_ws.data_ptr(), self._prefill_ws.numel(), st), "decoder_prefill")
            if tmp_logits is not None:
                sel = need_logits.to(dev)
                logits[sel] = tmp_logits[sel]
            done |= need_logits
        return logits

    def generate(self, prompts: List[List[int]], block_table, max_new_tokens: int, prefill_chunk: int = 512) -> List[List[int]]:
        """Greedy generation for a batch of prompts (tp = 1): real prefill of the prompts (large-M GEMMs, causal multi-row
        attention), greedy first token from the prefill logits, then graph-replayed decode steps."""
        B = len(prompts)
        if B > self.max_batch or any(len(pr) < 1 for pr in prompts):
            raise _C.Mi355Error("generate: batch exceeds max_batch or empty prompt")
        self.check_room([len(pr) for pr in prompts], max_new_tokens - 1, block_table, "generate")
        first = ops.argmax(self.prefill(prompts, block_table, chunk=prefill_chunk))
        out = [first.clone()]
        self.set_inputs([0] * B, [len(pr) for pr in prompts], block_table)
        self.token_ids[:B].copy_(first)
        self.capture(B)
        for _ in range(max_new_tokens - 1):
            self.replay(B, 1)
            out.append(self.token_ids[:B].clone())
        toks = torch.stack(out, 1).cpu().tolist()
        if self.oob_count():
            raise _C.Mi355Error("generate: the KV writer dropped out-of-range tokens (corrupt block table?)")
        return toks

    def generate_sampled(self, prompts: List[List[int]], block_table, max_new_tokens: int, *, top_k=0, top_p=1.0, temperature=1.0,
                         repetition_penalty=None, presence_penalty=None, frequency_penalty=None, no_repeat_ngram_size=None,
                         seed: int = 0, prefill_chunk: int = 512) -> List[List[int]]:
        """Generation through the sampler's non-greedy branch (tp = 1): prefill, then per token one step without sampling
        (logits only) and `sampler.sample_greedy` over them -- temperature, repetition / presence / frequency penalties over the
        tokens so far, no-repeat n-gram ban, top-k / top-p filter, draw.  Scalars or one value per sequence; the uniforms of the draw
        come from `seed` (one per sequence and token), so a run is reproducible.  top_k = 1 is greedy decoding (== generate())."""
        from . import sampler
        B = len(prompts)
        if B > self.max_batch or any(len(pr) < 1 for pr in prompts):
            raise _C.Mi355Error("generate_sampled: batch exceeds max_batch or empty prompt")
        self.check_room([len(pr) for pr in prompts], max_new_tokens - 1, block_table, "generate_sampled")
        vec = lambda v, dt: None if v is None else (torch.as_tensor(v, dtype=dt).reshape(-1).expand(B).contiguous() if torch.as_tensor(v).numel() == 1
                                                    else torch.as_tensor(v, dtype=dt).reshape(B))
        k, p_, t = vec(top_k, torch.int32), vec(top_p, torch.float32), vec(temperature, torch.float32)
        pen = repetition_penalty is not None or presence_penalty is not None or frequency_penalty is not None
        rep = vec(1.0 if repetition_penalty is None else repetition_penalty, torch.float32) if pen else None
        pres = vec(0.0 if presence_penalty is None else presence_penalty, torch.float32) if pen else None
        freq = vec(0.0 if frequency_penalty is None else frequency_penalty, torch.float32) if pen else None
        ngram = vec(no_repeat_ngram_size, torch.int32)
        lens = torch.tensor([len(pr) for pr in prompts], dtype=torch.int32)
        cols = int(lens.max()) + max_new_tokens + 1
        hist = torch.zeros(B, cols, dtype=torch.int32)            # row b: its prompt, then its generated tokens, compact
        for b, pr in enumerate(prompts):
            hist[b, : len(pr)] = torch.tensor(pr, dtype=torch.int32)
        seq = lens.clone()                                        # valid tokens per row
        u = torch.rand(max_new_tokens, B, generator=torch.Generator().manual_seed(seed))
        logits = self.prefill(prompts, block_table, chunk=prefill_chunk)
        self.set_inputs([0] * B, lens.tolist(), block_table)
        out = []
        for i in range(max_new_tokens):
            step = int(seq.max())                                 # column that receives the new token (CudaSampleOp.cc:797-799)
            params = sampler.GreedyParams(logits=logits, input_lengths=lens, sequence_lengths=seq, token_ids=hist[:, : step + 1].contiguous(),
                                          step=step, top_k=k, top_p=p_, temperature=t, repetition_penalty=rep, presence_penalty=pres,
                                          frequency_penalty=freq, no_repeat_ngram_size=ngram, uniform=u[i])
            ids = sampler.sample_greedy(params)
            idc = ids.cpu()
            hist[torch.arange(B), seq.long()] = idc
            seq += 1
            out.append(idc)
            if i + 1 == max_new_tokens:
                break
            self.token_ids[:B].copy_(ids)
            self.forward(B)                                       # logits of the next token; positions advance here, not in the step
            self.positions[:B] += 1
            logits = self.logits[:B]
        if self.oob_count():
            raise _C.Mi355Error("generate_sampled: the KV writer dropped out-of-range tokens (corrupt block table?)")
        return torch.stack(out, 1).tolist()

    def capture(self, B: int):
        _C.check(self.lib.mi355_decoder_capture(self.handle, B), "decoder_capture")

    def replay(self, B: int, nsteps: int = 1):
        _C.check(self.lib.mi355_decoder_replay(self.handle, B, nsteps, self._st()), "decoder_replay")

    def profile(self, B: int, nsteps: int):
        ms, n = (C.c_float * len(_C.KC_NAMES))(), (C.c_int32 * len(_C.KC_NAMES))()
        _C.check(self.lib.mi355_decoder_profile(self.handle, B, nsteps, ms, n, self._st()), "decoder_profile")
        return {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(_C.KC_NAMES)}

    def attach_allreduce(self, ar, vocab_offset: int):
        """tp > 1: run the all-reduce points inside the C++ step (fused split-K reduce + all-reduce + residual + norm,
        cross-rank greedy argmax): step() / capture() / replay() then drive the whole tensor-parallel step."""
        _C.check(self.lib.mi355_decoder_attach_allreduce(self.handle, ar.handle, int(vocab_offset)), "decoder_attach_allreduce")
        self._ar = ar

    def attach_collective(self, transport, vocab_offset: int):
        """tp > 1 without the IPC all-reduce: run the TP points through an external transport inside the C++ step
        (distributed.RcclTransport: ncclAllReduce / ncclAllGather on the step's stream, capturable) -- the reference's
        fallback under graph capture (rocm_rccl.py:511-572).  step() / capture() / replay() then work as with attach_allreduce."""
        _C.check(self.lib.mi355_decoder_attach_collective(self.handle, C.byref(transport.collective), int(vocab_offset)),
                 "decoder_attach_collective")
        self._transport = transport

    def set_weight_prefetch(self, mask: int):
        """Weight prefetch one launch ahead (mi355_decoder_set_weight_prefetch, _C.PF_* bits; 0 = off; the engine's default is
        _C.PF_QKV_IN_FOLD, the spare blocks of the slab fold reading the next QKV weights -- the side-stream bits are off).
        Captured graphs are dropped: capture again."""
        _C.check(self.lib.mi355_decoder_set_weight_prefetch(self.handle, int(mask)), "decoder_set_weight_prefetch")

    def set_embedding_split(self, on: bool = True):
        """The embedding tensor of this engine is the rank's [vocab, hidden / tp] column slice (split_embedding_tp): look it up and
        all-gather the hidden dimension every step (modules/base/common/embedding.py:50-58).  After attach_allreduce."""
        if on and self.embedding.shape[1] * self.tp_size != self.cfg.hidden:
            raise ValueError(f"embedding slice is {tuple(self.embedding.shape)}, expected [vocab, {self.cfg.hidden // self.tp_size}]")
        _C.check(self.lib.mi355_decoder_set_embedding_split(self.handle, 1 if on else 0), "decoder_set_embedding_split")

    # ---- tp > 1: the step cut at the all-reduce points (causal_attention.py:91-92, dense_mlp.py:104-105)
    def step_tp(self, B: int, sample: bool = True):
        if getattr(self, "_transport", None) is not None:
            raise _C.Mi355Error("step_tp: a transport is attached, the C++ step already all-reduces: use step()")
        st, h, lib = self._st(), self.handle, self.lib
        ar = self.ar_buf[:B]
        _C.check(lib.mi355_decoder_begin(h, B, st), "decoder_begin")
        for l in range(self.cfg.num_layers):
            _C.check(lib.mi355_decoder_layer_attn(h, l, st), "decoder_layer_attn")
            all_reduce(ar, Group.TP)
            _C.check(lib.mi355_decoder_layer_mlp(h, l, st), "decoder_layer_mlp")
            all_reduce(ar, Group.TP)
        _C.check(lib.mi355_decoder_finish(h, 0, st), "decoder_finish")
        if sample:   # vocab-split lm_head: gather logits, greedy on the full row (PyWrappedModel.cc:915-936)
            full = all_gather(self.logits[:B], Group.TP)
            self.token_ids[:B].copy_(ops.argmax(full.contiguous()))
            self.positions[:B] += 1
