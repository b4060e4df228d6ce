"""CPU oracle for the quantized decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under rtp_llm_amd/ may import this module; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
the checker / the timed CPU baseline — never as the product path.

It is a plain-torch (CPU, fp32 math) restatement of the reference's algorithm for
this path, each function citing the reference lines it follows.  Pinning status
(see DESIGN.md "Oracle"):
  * GPTQ/AWQ canonicalisation + INT8 autoquant: PINNED against golden vectors generated
    by importing the reference's own rtp_llm/device/device_impl.py (tests/golden/quant_*.npz,
    generator oracle/gen_golden.py).
  * RMSNorm / NeoX RoPE / paged decode attention (fp16 KV, and per-token-scaled 8-bit KV) / SiLU-gate MLP: PINNED against
    outputs of the reference's own torch reference implementations (RMSNormTorch, _torch_reference, run_native /
    ref_masked_attention, DenseMLP -- the functions its ROCm unit tests compare the native kernels with at
    atol=rtol=1e-2), executed unmodified by oracle/gen_golden.py -> tests/golden/ref_layers.npz, checked by
    tests/test_oracle_pinned.py (bit-equal, or within one fp16 ulp where the fp32 contraction order differs).
  * The same four in bf16 (the functions are dtype-generic: fp32 math, rounding to the input dtype where the reference's tensors
    have that dtype): PINNED against the same reference implementations executed on bf16 tensors ->
    tests/golden/ref_layers_bf16.npz (RMSNorm bit-equal; RoPE / attention within one bf16 ulp).
  * Scaled RoPE styles (linear / llama3 / yarn): PINNED bit for bit, see rope_inv_freq and oracle/gen_rope_golden.py.
  * Dynamic-NTK RoPE (rope_dynamic_ntk_bases): "dynamic" PINNED (2e-6 relative: fp32 here and on the device, double in the torch form) against
    the reference's own torch form of the style, DeepseekV3DynamicNTKScalingRotaryEmbedding (rtp_llm/models/rotary_embedding/
    deepseek_rotary_embedding.py:80-113 -> tests/golden/rope_styles.npz dynntk_*, tests/test_cpu_host.py); that the DECODE writer passes the
    cached length as seq_len is read off the device code (fused_rope_kvcache_kernel.cu:1341-1392).  "qwen_dynamic": PARITY UNPINNED --
    device code only in the reference, restated from the header.
  * Chain rejection sampling (speculative verify): PINNED against the reference's own known-answer kernel tests
    (bindings/cuda/test/CudaSpeculativeSamplingTest.cc:36-366, transcribed in tests/spec_vectors.py).
  * W4A16 / W8A16 GEMM results and INT8 KV-cache numerics: PARITY UNPINNED — the reference
    snapshot contains no kernel, test or golden vector for them; the oracle
    defines them from the loader formulas (W = scale*(q - z - gptq_flag), W = q*scale_col)
    and the FasterTransformer-lineage KV convention (scale = amax/127 per token*kv_head).
"""
import math
from typing import List, Optional, Tuple

import torch

# --------------------------------------------------------------------------- weights
def dequant_groupwise(q: torch.Tensor, z_eff: torch.Tensor, scales: torch.Tensor, group_size: int) -> torch.Tensor:
    """W[k,n] = scale[k//g, n] * (q[k,n] - z_eff[k//g, n]) in fp32 (exact: fp16 scale x small int).
    Reference formula: device_impl.py:283-289 (W = q_s*scale + (8 - z - GPTQ_FLAG)*scale with
    q_s = q - 8), i.e. W = scale*(q - z - GPTQ_FLAG)."""
    K, N = q.shape
    if K % group_size == 0:   # the same two fp32 operations per element, broadcast over the group instead of materialised (10x faster)
        w = q.reshape(K // group_size, group_size, N).float()
        w.sub_(z_eff.float().unsqueeze(1)).mul_(scales.float().unsqueeze(1))
        return w.view(K, N)
    s = scales.float().repeat_interleave(group_size, dim=0)[:K]
    z = z_eff.float().repeat_interleave(group_size, dim=0)[:K]
    return s * (q.float() - z)


def dequant_reference_folded(q: torch.Tensor, z_eff: torch.Tensor, scales: torch.Tensor, group_size: int) -> torch.Tensor:
    """The representation the reference hands its kernel: signed nibble * scale + fp16(zeros_x_scales)."""
    K, N = q.shape
    zs = ((8 - z_eff.to(torch.int16)).to(scales.dtype) * scales).half()  # device_impl.py:286-289
    s = scales.float().repeat_interleave(group_size, dim=0)[:K]
    zsr = zs.float().repeat_interleave(group_size, dim=0)[:K]
    return (q.float() - 8.0) * s + zsr


def dequant_int8(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """W = q * scale_col (a1, device_impl.py:183-192)."""
    return q.float() * scale.float().reshape(1, -1)


def linear(x: torch.Tensor, w_f32: torch.Tensor, bias: Optional[torch.Tensor] = None, out_f32: bool = False) -> torch.Tensor:
    """y = x @ W (+bias): fp32 accumulate, one rounding to the activation dtype
    (LinearBase contract, linear_base.py:75-85; SURVEY 8c' "GEMM")."""
    y = x.float() @ w_f32
    if bias is not None:
        y = y + bias.float()
    return y if out_f32 else y.to(x.dtype)


# --------------------------------------------------------------------------- layer math
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """modules/base/common/norm.py:83-92: fp32 normalise, cast to input dtype, then * weight."""
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    xn = (xf * torch.rsqrt(var + eps)).to(x.dtype)
    return weight * xn


def silu_mul(gate_up: torch.Tensor) -> torch.Tensor:
    """FusedSiluAndMul (modules/base/rocm/activation.py:9-24; ref hybrid/test/dense_mlp_ref.py:28-34):
    silu(gate) * up, gate = first half; fp32 internally, one rounding."""
    I = gate_up.shape[-1] // 2
    g, u = gate_up[..., :I].float(), gate_up[..., I:].float()
    return (torch.nn.functional.silu(g) * u).to(gate_up.dtype)


def rope_cos_sin(rope_dim: int, theta: float, max_pos: int, rope_scale: float = 1.0) -> torch.Tensor:
    """genBaseCache (cpp/model_utils/RopeCache.cc:16-41), interleaved {cos,sin} form:
    table[pos, i] = (cos, sin)(pos/scale * theta^(-2i/dim)), fp32."""
    inv_freq = 1.0 / torch.pow(torch.tensor(float(theta)), torch.arange(0, rope_dim, 2).float() / rope_dim)
    t = torch.arange(int(max_pos * rope_scale)).float() / rope_scale
    freqs = torch.outer(t, inv_freq)
    return torch.stack((freqs.cos(), freqs.sin()), dim=-1).contiguous()  # [max_pos, dim/2, 2]


def rope_inv_freq(rope_dim: int, theta: float, scaling: Optional[dict] = None) -> Tuple[torch.Tensor, float]:
    """Per-frequency angular step and cos/sin multiplier of the reference's RoPE styles, fp32
    (apply_rope dispatch, bindings/common/kernels/rotary_position_embedding.h:904-970; config mapping models/llama.py:87-117):
      * base / linear: inv_freq / scale                       LinearScaleRope, :355-364
      * llama3: wavelength-banded rescale                     Llama3Rope, :418-442
      * yarn: ramp between interpolated and extrapolated      YarnRope, :366-416 (correction range from beta_fast / beta_slow,
        original max positions; cos / sin scaled by mscale = 0.1 ln(factor) + 1, Llama.get_mscale models/llama.py:30-34)
    scaling: {"type", "factor", ...} with HF's key names.  The DynamicNTK styles have no single step per frequency (the base
    depends on the cached length): see rope_dynamic_ntk_bases."""
    idx = torch.arange(0, rope_dim, 2).float()
    inv = 1.0 / torch.pow(torch.tensor(float(theta)), idx / rope_dim)
    if not scaling:
        return inv, 1.0
    kind = scaling.get("rope_type", scaling.get("type"))
    factor = float(scaling.get("factor", 1.0))
    if kind in (None, "default"):
        return inv, 1.0
    if kind == "linear":
        return inv / factor, 1.0
    if kind == "llama3":
        old = float(scaling["original_max_position_embeddings"])
        lo_f, hi_f = float(scaling["low_freq_factor"]), float(scaling["high_freq_factor"])
        wavelen = 2 * math.pi / inv
        smooth = (old / wavelen - lo_f) / (hi_f - lo_f)
        mid = (1 - smooth) * inv / factor + smooth * inv
        out = torch.where(wavelen < old / hi_f, inv, torch.where(wavelen > old / lo_f, inv / factor, mid))
        return out, 1.0
    if kind == "yarn":
        orig = int(scaling["original_max_position_embeddings"])
        beta_fast, beta_slow = int(scaling.get("beta_fast", 32)), int(scaling.get("beta_slow", 1))
        extrapolation = float(scaling.get("extrapolation_factor", 1.0))
        corr = lambda rot: rope_dim * math.log(orig / (rot * 2 * math.pi)) / (2 * math.log(int(theta)))
        low, high = float(max(math.floor(corr(beta_fast)), 0)), float(min(math.ceil(corr(beta_slow)), rope_dim - 1))
        if low == high:
            high += 0.001
        ramp = torch.clamp((torch.arange(rope_dim // 2).float() - low) / (high - low), 0, 1)
        mask = (1 - ramp) * extrapolation
        out = (inv / factor) * (1 - mask) + inv * mask
        mscale = float(scaling.get("mscale_override", 0.1 * math.log(factor) + 1.0 if factor > 1 else 1.0))
        return out, mscale
    raise ValueError(f"rope scaling {kind!r}: no single step per frequency (dynamic NTK: rope_dynamic_ntk_bases)")


DYNAMIC_NTK_KINDS = ("dynamic", "qwen_dynamic")


def rope_dynamic_ntk_bases(rope_dim: int, theta: float, max_pos: int, scaling: dict) -> torch.Tensor:
    """Base of the rotation per DECODE position under the dynamic-NTK styles, fp32 [max_pos].  The reference's decode writer
    passes the number of tokens already cached -- the position p of the new token -- as `seq_len`
    (fused_rope_kvcache_kernel.cu:1341-1392: sequence_length = sequence_lengths[b], tlength = sequence_length), and apply_rope
    replaces the base once it exceeds the original context (rotary_position_embedding.h:925-951):
      * "dynamic"      (RopeStyle::DynamicNTK, :889-893): base (scale p / orig - (scale - 1)) ^ (dim / (dim - 2)), scale = factor
      * "qwen_dynamic" (RopeStyle::QwenDynamicNTK, :895-902): base (max(2 ^ ceil(log2(p / orig) + 1) - 1, 1)) ^ (dim / (dim - 2))
    so for decode the style IS a function of the position.  (A prompt longer than the original context rotates ALL its tokens with
    the base of the prompt length, context_rope: not a position-indexed table; the host refuses that case.)
    "dynamic" is pinned by the reference's torch form of the same formula (DeepseekV3DynamicNTKScalingRotaryEmbedding,
    deepseek_rotary_embedding.py:80-113; golden vectors dynntk_* of tests/golden/rope_styles.npz); "qwen_dynamic" is UNPINNED: restated from
    the header's device code, which cannot run here."""
    kind = scaling.get("rope_type", scaling.get("type"))
    orig = int(scaling["original_max_position_embeddings"])
    p = torch.arange(max_pos, dtype=torch.float32)
    expo = torch.tensor(float(rope_dim) / (rope_dim - 2.0), dtype=torch.float32)
    base = torch.full((max_pos,), float(theta), dtype=torch.float32)
    if kind == "dynamic":
        scale = torch.tensor(float(scaling.get("factor", 1.0)), dtype=torch.float32)
        grown = base * torch.pow(scale * p / float(orig) - (scale - 1.0), expo)
    elif kind == "qwen_dynamic":
        ctx = torch.log(p.clamp(min=1.0) / float(orig)) / math.log(2.0) + 1.0
        ntk = torch.clamp(torch.pow(torch.tensor(2.0), torch.ceil(ctx)) - 1.0, min=1.0)
        grown = base * torch.pow(ntk, expo)
    else:
        raise ValueError(f"not a dynamic-NTK style: {kind!r}")
    return torch.where(p > float(orig), grown, base)


def rope_cos_sin_scaled(rope_dim: int, theta: float, max_pos: int, scaling: Optional[dict] = None) -> torch.Tensor:
    """{cos, sin} table [max_pos][dim/2][2] of a scaled style: angle = pos * rope_inv_freq, both scaled by mscale
    (normal_rope + sin_cos_scale(), rotary_position_embedding.h:350-416).  Dynamic-NTK styles: row p with the base of position p
    (rope_dynamic_ntk_bases, DefaultRope: angle = p / base_p ^ (2 i / dim))."""
    if scaling and scaling.get("rope_type", scaling.get("type")) in DYNAMIC_NTK_KINDS:
        bases = rope_dynamic_ntk_bases(rope_dim, theta, max_pos, scaling)
        idx = torch.arange(0, rope_dim, 2).float() / rope_dim
        freqs = torch.arange(max_pos).float()[:, None] / torch.pow(bases[:, None], idx[None, :])
        return torch.stack((freqs.cos(), freqs.sin()), dim=-1).contiguous()
    inv, mscale = rope_inv_freq(rope_dim, theta, scaling)
    freqs = torch.outer(torch.arange(max_pos).float(), inv)
    return torch.stack((freqs.cos() * mscale, freqs.sin() * mscale), dim=-1).contiguous()


def apply_rope(x: torch.Tensor, positions: torch.Tensor, cos_sin: torch.Tensor) -> torch.Tensor:
    """NeoX split-halves RoPE (rotary_position_embedding.h:444-449,481-505; torch restatement
    test_fused_qkv_transpose_v3.py:246-307): x' = cos*x - sin*y, y' = cos*y + sin*x in fp32,
    one rounding.  x: [T, H, D]; positions: [T]."""
    T, H, D = x.shape
    half = D // 2
    cs = cos_sin[positions.long()]           # [T, half, 2]
    cos, sin = cs[..., 0].unsqueeze(1), cs[..., 1].unsqueeze(1)
    xf = x.float()
    lo, hi = xf[..., :half], xf[..., half:]
    out = torch.cat((lo * cos - hi * sin, hi * cos + lo * sin), dim=-1)
    return out.to(x.dtype)


def quant_kv_int8(x: torch.Tensor):
    """INT8 KV convention (inferred): per (token, kv head) scale = amax/127 (fp32),
    q = clamp(rint(x/scale), -128, 127) — rounding of rocm_utils/_cast_to_int8.h:8-14.
    x: [..., hd] fp16.  Returns (int8, fp32 scale[...])."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1)
    scale = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
    q = torch.clamp(torch.round(xf / scale.unsqueeze(-1)), -128, 127).to(torch.int8)
    return q, scale


def attention_decode(q: torch.Tensor, keys: torch.Tensor, values: torch.Tensor, scale: float,
                     k_scale: Optional[torch.Tensor] = None, v_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One sequence, q_len = 1 — ref_masked_attention (modules/base/rocm/test/rocm_fmha_test.py:262-298):
    fp32 logits scale*q.k (* k_scale), fp32 softmax, (* v_scale), P.V, cast.
    q [nh, hd]; keys/values [ctx, nkv, hd] (fp16 or int8); k_scale/v_scale [ctx, nkv] fp32."""
    nh, hd = q.shape
    ctx, nkv, _ = keys.shape
    g = nh // nkv
    # head h reads kv head h // g: the g query heads of a kv head are one batched product against its [ctx, hd] keys / values
    # (the same sums as the reference's repeat_interleave form, 35x less host memory traffic at ctx 4096)
    qg = q.float().view(nkv, g, hd)
    kf = keys.float().permute(1, 0, 2)                 # [nkv, ctx, hd]
    vf = values.float().permute(1, 0, 2)
    logits = scale * torch.bmm(qg, kf.transpose(1, 2))  # [nkv, g, ctx]
    if k_scale is not None:
        logits = logits * k_scale.float().t().unsqueeze(1)
    p = torch.softmax(logits, dim=-1)
    if v_scale is not None:
        p = p * v_scale.float().t().unsqueeze(1)
    out = torch.bmm(p, vf)                             # [nkv, g, hd]
    return out.reshape(nh, hd).to(q.dtype)


def greedy(logits_f32: torch.Tensor) -> torch.Tensor:
    """top_k == 1 fast path: argmax on fp32 logits, no softmax (bindings/core/CudaSampleOp.cc:687-700)."""
    return torch.argmax(logits_f32, dim=-1).to(torch.int32)


# --------------------------------------------------------------------------- whole decoder
class OracleKV:
    """Natural-layout KV store per layer: lists of [nkv, hd] rows per sequence (fp16 or int8+scale)."""

    def __init__(self, num_layers: int, batch: int, int8: bool, forced=None):
        """forced (INT8 only, test plumbing): callable (layer, sequence, token index) -> (k_codes [nkv,hd] int8, k_scale [nkv],
        v_codes, v_scale) or None.  When it returns codes, THEY are stored instead of this store's own quantisation of (k, v)
        -- the attention that follows then reads exactly the int8 codes the kernel under test wrote, so a code flipped by a
        1-ulp fp16 difference upstream cannot contribute to the comparison; the flips are counted in `flips` / `codes`."""
        self.int8 = int8
        self.forced = forced
        self.flips = self.codes = self.max_delta = 0
        self.max_scale_rel = 0.0      # largest relative difference between a forced scale and this store's own
        self.k = [[[] for _ in range(batch)] for _ in range(num_layers)]
        self.v = [[[] for _ in range(batch)] for _ in range(num_layers)]
        self.ks = [[[] for _ in range(batch)] for _ in range(num_layers)]
        self.vs = [[[] for _ in range(batch)] for _ in range(num_layers)]

    def append(self, layer: int, b: int, k: torch.Tensor, v: torch.Tensor):
        if self.int8:
            kq, ksc = quant_kv_int8(k); vq, vsc = quant_kv_int8(v)
            f = self.forced(layer, b, len(self.k[layer][b])) if self.forced is not None else None
            if f is not None:
                fk, fks, fv, fvs = f
                for own, got in ((kq, fk), (vq, fv)):
                    d = (own.int() - got.int()).abs()
                    self.flips += int((d > 0).sum()); self.codes += d.numel(); self.max_delta = max(self.max_delta, int(d.max()))
                # the scales the kernel wrote are compared with this store's own (amax / 127 of nearly the same fp16 row): a wrong scale
                # plane would otherwise be read by both sides of the comparison and cancel out
                for own, got in ((ksc, fks), (vsc, fvs)):
                    rel = ((own.float() - got.float()).abs() / own.float().abs().clamp_min(1e-12)).max()
                    self.max_scale_rel = max(self.max_scale_rel, float(rel))
                kq, ksc, vq, vsc = fk.to(kq.dtype), fks.to(ksc.dtype), fv.to(vq.dtype), fvs.to(vsc.dtype)
            self.k[layer][b].append(kq); self.v[layer][b].append(vq)
            self.ks[layer][b].append(ksc); self.vs[layer][b].append(vsc)
        else:
            self.k[layer][b].append(k); self.v[layer][b].append(v)

    def get(self, layer: int, b: int):
        K, V = torch.stack(self.k[layer][b]), torch.stack(self.v[layer][b])
        if self.int8:
            return K, V, torch.stack(self.ks[layer][b]), torch.stack(self.vs[layer][b])
        return K, V, None, None


class OracleDecoder:
    """Qwen2/Llama decoder in fp32-accumulate torch: embedding -> N x [RMSNorm, QKV(+bias), RoPE,
    KV append, attention, O, +res, RMSNorm, gate_up, SiLU*mul, down, +res] -> RMSNorm -> lm_head
    -> argmax (Qwen3DecoderLayer.forward, models_py/model_desc/qwen3.py:57-79,124-138;
    CausalAttention causal_attention.py:75-93; DenseMLP dense_mlp.py:95-106; post-layers
    PyWrappedModel.cc:984-1047).  Weights are given dequantised ([K,N] fp32)."""

    def __init__(self, cfg: dict, weights: dict):
        self.cfg, self.w = cfg, weights
        self.cos_sin = (rope_cos_sin_scaled(cfg["hd"], cfg["rope_theta"], cfg["max_pos"], cfg["rope_scaling"]) if cfg.get("rope_scaling")
                        else rope_cos_sin(cfg["hd"], cfg["rope_theta"], cfg["max_pos"]))

    def forward_tokens(self, token_ids: torch.Tensor, positions: torch.Tensor, kv: OracleKV, seq_idx: List[int], trace: Optional[list] = None):
        """Process T tokens (token t belongs to sequence seq_idx[t] at position positions[t]); the
        tokens of one sequence must be given in order.  Returns (hidden fp16 [T,H], logits fp32 [T,V]).
        trace (test plumbing): a list that receives the residual stream after every half layer -- (layer, "attn" | "mlp", h [T,H]) --
        so that a full-depth run can show where a difference starts instead of only that the logits differ."""
        c, w = self.cfg, self.w
        nh, nkv, hd, eps = c["nh"], c["nkv"], c["hd"], c["rms_eps"]
        h = w["embedding"][token_ids.long()]
        T = h.shape[0]
        for l in range(c["num_layers"]):
            L = w["layers"][l]
            x = rmsnorm(h, L["input_norm"], eps)
            qkv = linear(x, L["qkv"], L.get("qkv_bias"))
            qh = qkv[:, : nh * hd].reshape(T, nh, hd)
            kh = qkv[:, nh * hd: (nh + nkv) * hd].reshape(T, nkv, hd)
            vh = qkv[:, (nh + nkv) * hd:].reshape(T, nkv, hd)
            qh, kh = apply_rope(qh, positions, self.cos_sin), apply_rope(kh, positions, self.cos_sin)
            attn = torch.empty(T, nh * hd, dtype=h.dtype)
            for t in range(T):
                b = seq_idx[t]
                kv.append(l, b, kh[t], vh[t])
                K, V, ks, vs = kv.get(l, b)
                attn[t] = attention_decode(qh[t], K, V, 1.0 / math.sqrt(hd), ks, vs).reshape(-1)
            o = linear(attn, L["o"])
            h = h + o
            if trace is not None:
                trace.append((l, "attn", h.clone()))
            x = rmsnorm(h, L["post_norm"], eps)
            act = silu_mul(linear(x, L["gate_up"]))
            h = h + linear(act, L["down"])
            if trace is not None:
                trace.append((l, "mlp", h.clone()))
        hn = rmsnorm(h, w["final_norm"], eps)
        logits = linear(hn, w["lm_head"], out_f32=True)
        return hn, logits


# --------------------------------------------------------------------------- speculative verify
def rejection_sample(draft_token_ids, target_token_ids, target_probs, uniform_samples, do_sample, draft_probs=None):
    """CPU restatement of rejection_sampling_kernel (models_py/bindings/rocm/speculative_sampling/sampling.cu:306-475),
    plain Python loops (small cases only).  draft_token_ids [B,g], target_token_ids [B,g+1], target_probs [B,g+1,V],
    uniform_samples [B,g+1], do_sample [B], draft_probs [B,g,V] or None (point mass at the draft token).
    Returns (output_token_ids [B,g+1] padded with -1, accepted [B] = accepted drafts + 1).  The residual draw is the
    first index whose inclusive fp32 prefix sum of relu(q - p), taken in index order, exceeds u * sum (sampling.cu:421-463)."""
    import numpy as np
    d = np.asarray(draft_token_ids); t = np.asarray(target_token_ids)
    q = np.asarray(target_probs, dtype=np.float32); u = np.asarray(uniform_samples, dtype=np.float32)
    B, G = d.shape
    V = q.shape[-1]
    out = np.full((B, G + 1), -1, dtype=np.int32)
    acc = np.zeros(B, dtype=np.int32)
    for b in range(B):
        sample = bool(do_sample[b])
        pos, fallback = G, False
        for i in range(G):
            qi = q[b, i, d[b, i]]
            pi = np.float32(1.0) if draft_probs is None else np.float32(draft_probs[b][i][d[b, i]])
            accept = (np.float32(u[b, i] * pi) < qi) if sample else (t[b, i] == d[b, i])   # sampling.cu:352
            if accept:
                out[b, i] = d[b, i]
            else:
                pos = i
                if not sample:
                    out[b, i] = t[b, i]
                    fallback = True
                break
        acc[b] = pos + 1
        if pos == G:
            out[b, G] = t[b, G]
        if fallback or pos == G:
            continue
        p = np.zeros(V, dtype=np.float32)
        if draft_probs is None:
            p[d[b, pos]] = 1.0
        else:
            p = np.asarray(draft_probs[b][pos], dtype=np.float32)
        r = np.maximum(q[b, pos] - p, np.float32(0))
        total = np.float32(0)
        for x in r:
            total = np.float32(total + x)
        thr = np.float32(u[b, min(pos + 1, G)] * total)
        c, found = np.float32(0), V - 1
        for j in range(V):
            c = np.float32(c + r[j])
            if r[j] > 0 and c > thr:
                found = j
                break
        out[b, pos] = found
    return torch.from_numpy(out), torch.from_numpy(acc)


def sample_rows(probs, uniform):
    """ids[r] = first index whose inclusive fp32 prefix sum of probs[r] (index order) exceeds uniform[r] * sum(probs[r]) --
    sampling from the probabilities (bindings/core/CudaSampleOp.cc:702-737, top_k = 0 / top_p = 1 branch)."""
    import numpy as np
    q = np.asarray(probs, dtype=np.float32); u = np.asarray(uniform, dtype=np.float32)
    out = np.zeros(q.shape[0], dtype=np.int32)
    for r in range(q.shape[0]):
        total = np.float32(0)
        for x in q[r]:
            total = np.float32(total + max(x, np.float32(0)))
        thr, c, found = np.float32(u[r] * total), np.float32(0), q.shape[1] - 1
        for j, x in enumerate(q[r]):
            x = max(x, np.float32(0))
            c = np.float32(c + x)
            if x > 0 and c > thr:
                found = j
                break
        out[r] = found
    return torch.from_numpy(out)


def softmax_rows(logits: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
    """softmax(logits / T) per row in fp32 (the distribution handed to rejection sampling)."""
    return torch.softmax(logits.float() / temperature, dim=-1)


def apply_penalties(logits, temperature=None, repetition_penalty=None, presence_penalty=None, frequency_penalty=None,
                    output_ids=None, input_lengths=None, max_input_length=0, step=0):
    """fp32, returns a new tensor.  Temperature as batchApplyTemperaturePenalty
    (bindings/common/kernels/sampling_penalty_kernels.cu:26-54): logit * (1 / (T + 1e-6)); then batchApplyPenaltyLongSeq
    (:129-180): ids counted over output_ids[0:step, row] (positions [input_length, max_input_length) skipped, ids outside the
    vocabulary skipped), each seen id penalised once with its count."""
    import numpy as np
    x = (logits.detach().cpu().numpy() if isinstance(logits, torch.Tensor) else np.asarray(logits)).astype(np.float32, copy=True)
    B, V = x.shape
    if temperature is not None:
        t = np.asarray(temperature, dtype=np.float32)
        inv = (np.float32(1.0) / (t + np.float32(1e-6))).astype(np.float32)
        x = (x * inv[:, None]).astype(np.float32)
    if repetition_penalty is not None or presence_penalty is not None or frequency_penalty is not None:
        ids = np.asarray(output_ids)
        for b in range(B):
            n_in = int(input_lengths[b]) if input_lengths is not None else max_input_length
            count = np.zeros(V, dtype=np.int64)
            for index in range(step):
                if n_in <= index < max_input_length:
                    continue
                tok = int(ids[index, b])
                if 0 <= tok < V:
                    count[tok] += 1
            seen = count > 0
            row = x[b]
            if repetition_penalty is not None:
                r = np.float32(repetition_penalty[b])
                row[seen] = np.where(row[seen] < 0, row[seen] * r, row[seen] / r).astype(np.float32)
            if presence_penalty is not None:
                row[seen] = (row[seen] - np.float32(presence_penalty[b])).astype(np.float32)
            if frequency_penalty is not None:
                row[seen] = (row[seen] - (np.float32(frequency_penalty[b]) * count[seen].astype(np.float32)).astype(np.float32)).astype(np.float32)
    return torch.from_numpy(x)


def top_k_top_p_filter(probs, top_k, top_p):
    """The filter of the sampler's ROCm branch, statement for statement in torch (bindings/core/CudaSampleOp.cc:748-781):
    per row topk(k) -> entries below the k-th value zeroed; sort descending, cumsum, entries whose preceding mass exceeds p
    zeroed, scattered back; renormalised by clamp_min(sum, 1e-10).  Equal values are taken in index order (stable sort): the order
    the reference's known answers imply (CudaSamplerTest.cc:659-719 accepts token 1 of the tied {1, 2, 6}, not 2 or 6)."""
    out = probs.clone().float()
    B, V = out.shape
    for b in range(B):
        k = int(top_k[b]) if top_k is not None else 0
        k = V if k <= 0 else k
        if k < V:
            row = out[b]
            vals, _ = row.topk(k)
            row.masked_fill_(row < vals[-1], 0.0)
    for b in range(B):
        p = float(top_p[b]) if top_p is not None else 1.0
        if abs(p) < 1e-7:
            p = 1.0
        if abs(p - 1.0) >= 1e-7:
            row = out[b]
            sp, si = row.sort(dim=0, descending=True, stable=True)
            cs = sp.cumsum(0)
            sp = sp.masked_fill(cs - sp > p, 0.0)
            row.scatter_(0, si, sp)
    sums = out.sum(-1, keepdim=True)
    return out / sums.clamp_min(1e-10)


def ban_repeat_ngram(logits, token_ids, sequence_last_index, no_repeat_ngram_size):
    """fp32, returns a new tensor.  ban_repeat_ngram for beam_width = 1 (bindings/common/kernels/banRepeatNgram.cu:60-136): with
    N = last_index + 1 tokens and n = ngram size (0, or N < n: nothing), every i <= N - n with tokens[i : i + n - 1] equal to the
    last n - 1 tokens bans tokens[i + n - 1]."""
    out = logits.clone().float()
    for b in range(token_ids.shape[0]):
        n, N = int(no_repeat_ngram_size[b]), int(sequence_last_index[b]) + 1
        if n == 0 or N < n:
            continue
        t = [int(v) for v in token_ids[b][:N]]
        last = t[N - n + 1:N]
        for i in range(N - n + 1):
            if t[i:i + n - 1] == last:
                out[b, t[i + n - 1]] = float("-inf")
    return out
