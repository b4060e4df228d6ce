"""Golden vectors for the scaled RoPE styles.  TEST INFRASTRUCTURE ONLY; runs in the authoring container (needs
/root/reference and the offline `transformers` wheel); the committed output tests/golden/rope_styles.npz is what travels.

  * yarn:   the reference's own torch restatement, executed unmodified with its rtp_llm imports stubbed:
            DeepseekV3YarnRotaryEmbedding._set_cos_sin_cache + yarn_find_correction_range / yarn_linear_ramp_mask /
            yarn_get_mscale, rtp_llm/models_py/modules/hybrid/test/mla_attention_ref.py:58-160
            (the in-kernel form is YarnRope, bindings/common/kernels/rotary_position_embedding.h:366-416).
  * llama3: the reference has no torch restatement of Llama3Rope (rotary_position_embedding.h:418-442, config mapping
            rtp_llm/models/llama.py:108-116); the published formula it implements is transformers' (5.15.0)
            modeling_rope_utils._compute_llama3_parameters, executed here for Llama-3.1's shipped rope_scaling.

  * dynamic NTK ("dynamic"): the reference's own torch form of the style, DeepseekV3DynamicNTKScalingRotaryEmbedding
            (rtp_llm/models/rotary_embedding/deepseek_rotary_embedding.py:80-113; in-kernel form DynamicNTK,
            rotary_position_embedding.h:889-893), executed unmodified for a range of lengths past the original context: the
            inverse frequencies of each length and the table row of its last position.  (QwenDynamicNTK, :895-902, has no torch form
            in the reference: it stays unpinned.)

    python oracle/gen_rope_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, REF, _Stub  # noqa: E402


def main():
    for name in ["rtp_llm", "rtp_llm.config", "rtp_llm.config.quant_config", "rtp_llm.models_py", "rtp_llm.models_py.modules",
                 "rtp_llm.models_py.modules.base", "rtp_llm.models_py.modules.base.common", "rtp_llm.models_py.modules.base.common.norm",
                 "rtp_llm.models_py.modules.factory", "rtp_llm.ops", "rtp_llm.utils", "rtp_llm.utils.model_weight"]:
        sys.modules[name] = _Stub(name)
    spec = importlib.util.spec_from_file_location("mla_attention_ref", os.path.join(REF, "rtp_llm/models_py/modules/hybrid/test/mla_attention_ref.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    # yarn: Qwen2.5-style long-context setting (factor 4 over 32768) at head_dim 128, and a small case
    for tag, (dim, base, factor, orig, npos) in {"yarn_a": (128, 1000000, 4.0, 32768, 96), "yarn_b": (64, 10000, 8.0, 4096, 64)}.items():
        m = ref.DeepseekV3YarnRotaryEmbedding(dim, max_position_embeddings=npos, base=base, scaling_factor=factor,
                                              original_max_position_embeddings=orig, beta_fast=32, beta_slow=1, mscale=1, mscale_all_dim=0)
        m._set_cos_sin_cache(npos, "cpu", torch.float32)
        out[tag + "_cfg"] = np.array([dim, base, factor, orig, npos], dtype=np.float64)
        out[tag + "_inv_freq"] = m.inv_freq.numpy()
        out[tag + "_cos"] = m.cos_cached[:, :dim // 2].numpy()
        out[tag + "_sin"] = m.sin_cached[:, :dim // 2].numpy()
        out[tag + "_mscale"] = np.array([ref.yarn_get_mscale(factor, 1)], dtype=np.float64)
    # llama3: Llama-3.1's config.json rope_scaling
    from transformers import LlamaConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    rs = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
    for tag, (hd, theta) in {"llama3_a": (128, 500000.0), "llama3_b": (64, 10000.0)}.items():
        cfg = LlamaConfig(hidden_size=hd * 4, num_attention_heads=4, head_dim=hd, rope_theta=theta, rope_scaling=dict(rs),
                          max_position_embeddings=131072)
        inv, att = ROPE_INIT_FUNCTIONS["llama3"](cfg, "cpu")
        assert att == 1.0
        out[tag + "_cfg"] = np.array([hd, theta, rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"], rs["original_max_position_embeddings"]], dtype=np.float64)
        out[tag + "_inv_freq"] = inv.float().numpy()
    # dynamic NTK: the class rebuilds inv_freq from the length it is asked for (base in double precision, the rest fp32)
    spec = importlib.util.spec_from_file_location("deepseek_rotary_embedding", os.path.join(REF, "rtp_llm/models/rotary_embedding/deepseek_rotary_embedding.py"))
    dre = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dre)
    for tag, (dim, base, factor, orig) in {"dynntk_a": (128, 1000000, 4.0, 32), "dynntk_b": (64, 10000, 2.0, 16)}.items():
        lens = [orig - 3, orig, orig + 1, orig + 2, orig + 7, 2 * orig, 3 * orig + 5, 8 * orig]
        inv, cos, sin = [], [], []
        for S in lens:
            m = dre.DeepseekV3DynamicNTKScalingRotaryEmbedding(dim, max_position_embeddings=orig, base=base, scaling_factor=factor)
            m._set_cos_sin_cache(S, "cpu", torch.float32)
            inv.append(m.inv_freq.numpy().copy()); cos.append(m.cos_cached[S - 1, :dim // 2].numpy().copy()); sin.append(m.sin_cached[S - 1, :dim // 2].numpy().copy())
        out[tag + "_cfg"] = np.array([dim, base, factor, orig], dtype=np.float64)
        out[tag + "_lens"] = np.array(lens, dtype=np.int64)
        out[tag + "_inv_freq"] = np.stack(inv); out[tag + "_cos_last"] = np.stack(cos); out[tag + "_sin_last"] = np.stack(sin)
    np.savez(os.path.join(OUT, "rope_styles.npz"), **out)
    print("wrote", os.path.join(OUT, "rope_styles.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
