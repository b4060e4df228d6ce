"""CPU-side tests: the C-ABI library loads and exports every symbol include/mi355_decode.h declares
(no compute calls without a GPU), host logic (planning, TP split, module plumbing), oracle
self-consistency.  Runs with -m "not gpu"."""
import ctypes
import math
import os
import re

import pytest
import torch

from oracle import oracle
from rtp_llm_amd import _C, model, quant
from rtp_llm_amd.linear import LinearFactory, Mi355W4A16Linear

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mi355_decode.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rtp_llm_amd.build import build
    build(verbose=False)                      # hipcc cross-compiles gfx950 without a GPU
    lib = ctypes.CDLL(_C.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mi355_decode.h but not exported"
    assert set(declared) == set(_C.SIGNATURES), "ctypes SIGNATURES out of sync with the header"
    assert _C.lib().mi355_abi_version() == _C.ABI_VERSION


def test_argument_errors_are_reported_without_a_gpu():
    """Argument validation happens on the host before any launch: status + message, no exception across the ABI."""
    l = _C.lib()
    w = _C.Weight(None, None, 4, 256, 64, 256, 64, 128)
    assert l.mi355_linear_forward(None, 1, ctypes.byref(w), None, None, 0, None, 0, None) == _C.ERR_ARG
    assert b"null weight" in l.mi355_last_error()
    kv = _C.KVLayer(1, None, _C.KV_INT8, 16, 4, 96, 8)
    assert l.mi355_paged_decode_attn(1, ctypes.byref(kv), 1, 4, 1, 1, 28, 0.1, 64, 1, None, 0, None) == _C.ERR_ARG
    with pytest.raises(_C.Mi355Error):
        _C.check(_C.ERR_ARG, "x")
    assert l.mi355_paged_attn_workspace_bytes(64, 28, 128, 4096) == 64 * 28 * 32 * 130 * 4


def test_ops_refuse_cpu_tensors():
    from rtp_llm_amd import ops
    with pytest.raises(_C.Mi355Error):
        ops.rmsnorm(torch.zeros(2, 64, dtype=torch.float16), torch.ones(64, dtype=torch.float16), 1e-6)


def test_linear_factory_unique_dispatch():
    q = torch.randint(0, 16, (256, 32), dtype=torch.uint8)
    s = torch.rand(2, 32).half()
    z = torch.randint(0, 16, (2, 32), dtype=torch.uint8)
    lin = LinearFactory.create_linear(q, None, s, None, z)
    assert isinstance(lin, Mi355W4A16Linear) and lin.packed.group_size == 128 and lin.packed.wbits == 4
    with pytest.raises(ValueError):           # bf16 weights match no strategy, as in the reference factory
        LinearFactory.create_linear(torch.zeros(8, 8, dtype=torch.bfloat16))


def test_tp_split_is_exact_partition():
    """Column-parallel + row-parallel split of the canonical tensors reproduces the unsplit layer:
    sum over ranks of the oracle MLP / attention-projection outputs == unsplit output."""
    cfg = model.ModelConfig("t", 1, 256, 8, 4, 64, 512, 64)   # o rows 512/4 and inter 512/4 stay group(128)-aligned
    gen = torch.Generator().manual_seed(0)
    L = model.synth_layer(cfg, "w4", "cpu", gen)
    dense = lambda c: oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size)
    x = (torch.randn(3, 256, generator=gen) * 0.5)
    full_mlp = (torch.nn.functional.silu((x @ dense(L["gate_up"]))[:, :512]) * (x @ dense(L["gate_up"]))[:, 512:]) @ dense(L["down"])
    full_qkv = x @ dense(L["qkv"]) + L["qkv_bias"].float()
    for tp in (2, 4):
        acc = torch.zeros_like(full_mlp)
        for r in range(tp):
            Lr = model.split_layer_tp(L, cfg, tp, r)
            gu = x @ dense(Lr["gate_up"])
            Ir = 512 // tp
            acc += (torch.nn.functional.silu(gu[:, :Ir]) * gu[:, Ir:]) @ dense(Lr["down"])
            c = cfg.per_rank(tp)
            qkv_r = x @ dense(Lr["qkv"]) + Lr["qkv_bias"].float()
            nhr, nkr, hd = c.nh, c.nkv, c.hd
            assert torch.equal(qkv_r[:, : nhr * hd], full_qkv[:, r * nhr * hd:(r + 1) * nhr * hd])
            k0 = cfg.nh * hd + r * nkr * hd
            assert torch.equal(qkv_r[:, nhr * hd:(nhr + nkr) * hd], full_qkv[:, k0:k0 + nkr * hd])
        assert torch.allclose(acc, full_mlp, atol=1e-4, rtol=1e-4)


def test_oracle_attention_matches_plain_softmax_and_int8_roundtrip():
    g = torch.Generator().manual_seed(1)
    q = torch.randn(8, 64, generator=g).half()
    K, V = torch.randn(37, 2, 64, generator=g).half(), torch.randn(37, 2, 64, generator=g).half()
    out = oracle.attention_decode(q, K, V, 0.125)
    for h in range(8):
        p = torch.softmax(0.125 * (K[:, h // 4].float() @ q[h].float()), 0)
        assert torch.allclose(out[h].float(), p @ V[:, h // 4].float(), atol=2e-3)
    Kq, ks = oracle.quant_kv_int8(K)
    assert Kq.dtype == torch.int8 and int(Kq.abs().max()) == 127
    assert torch.allclose(Kq.float() * ks.unsqueeze(-1), K.float(), atol=float(ks.max()) * 0.51)


def test_oracle_rope_is_a_rotation_and_matches_reference_formula():
    cs = oracle.rope_cos_sin(64, 1e6, 128)
    x = torch.randn(5, 3, 64).half()
    pos = torch.tensor([0, 1, 17, 100, 127])
    y = oracle.apply_rope(x, pos, cs)
    assert torch.equal(y[0], x[0])                                    # position 0: identity
    assert torch.allclose(y.float().norm(dim=-1), x.float().norm(dim=-1), rtol=2e-3)
    # test_fused_qkv_transpose_v3.py:279-284 formula
    inv = 1e6 ** (-2.0 * torch.arange(32).float() / 64)
    ang = pos.float().unsqueeze(1) * inv
    lo, hi = x.float()[..., :32], x.float()[..., 32:]
    ref = torch.cat((lo * ang.cos().unsqueeze(1) - hi * ang.sin().unsqueeze(1), hi * ang.cos().unsqueeze(1) + lo * ang.sin().unsqueeze(1)), -1)
    assert torch.allclose(y.float(), ref, atol=4e-3)


def test_kvcache_layout_roundtrip_cpu():
    from rtp_llm_amd import kvcache
    kv, sc = kvcache.alloc_layer_cache(8, 2, 16, 64, True, "cpu")
    K = torch.randint(-128, 128, (40, 2, 64)).to(torch.int8)
    V = torch.randint(-128, 128, (40, 2, 64)).to(torch.int8)
    ks, vs = torch.rand(40, 2), torch.rand(40, 2)
    bt = torch.tensor([5, 2, 7, 0], dtype=torch.int32)
    kvcache.write_tokens(kv, sc, bt, 0, K, V, ks, vs)
    K2, V2, ks2, vs2 = kvcache.read_tokens(kv, sc, bt, 40)
    assert torch.equal(K2, K) and torch.equal(V2, V) and torch.equal(ks2, ks) and torch.equal(vs2, vs)
    # native intra-block layout: K block [nkv][page][hd], V block [nkv][hd][page]; 8-bit codes stored with the top bit flipped
    flip = lambda x: int(x) ^ -128
    assert kv[5, 0].reshape(2, 16, 64)[1, 3, 10] == flip(K[3, 1, 10])
    assert kv[5, 1].reshape(2, 64, 16)[1, 10, 3] == flip(V[3, 1, 10])
    assert kv[2, 1].reshape(2, 64, 16)[0, 7, 4] == flip(V[16 + 4, 0, 7])


def test_kv_head_replication_when_tp_exceeds_kv_heads():
    """Qwen2-7B (nkv = 4) at tp = 8: one kv head per rank, shared by the two ranks that own its query heads
    (get_sp_tensor splits K/V by gcd(nkv, tp), utils/model_weight.py:447-466)."""
    from rtp_llm_amd import model
    cfg = model.ModelConfig("t", 1, 256, 8, 2, 32, 512, 1024)
    assert cfg.per_rank(4).nkv == 1 and cfg.per_rank(4).nh == 2 and cfg.per_rank(2).nkv == 1
    L = model.synth_layer(cfg, "fp16", "cpu", torch.Generator().manual_seed(0))
    W, b = L["qkv"].w, L["qkv_bias"]
    for rank in range(4):
        S = model.split_layer_tp(L, cfg, 4, rank)
        kvh = rank // 2                                              # ranks (0,1) share kv head 0, (2,3) kv head 1
        q = W[:, rank * 64:(rank + 1) * 64]
        k = W[:, 256 + kvh * 32: 256 + (kvh + 1) * 32]
        v = W[:, 320 + kvh * 32: 320 + (kvh + 1) * 32]
        assert torch.equal(S["qkv"].w, torch.cat([q, k, v], 1))
        assert torch.equal(S["qkv_bias"], torch.cat([b[rank * 64:(rank + 1) * 64], b[256 + kvh * 32: 256 + (kvh + 1) * 32],
                                                     b[320 + kvh * 32: 320 + (kvh + 1) * 32]]))
    with pytest.raises(ValueError):
        model.ModelConfig("t", 1, 256, 12, 3, 32, 512, 1024).per_rank(2)   # 3 kv heads over 2 ranks: neither divides


def test_oracle_sample_rows_is_inverse_cdf():
    from oracle import oracle
    p = torch.tensor([[0.25, 0.0, 0.5, 0.25], [0.0, 0.0, 1.0, 0.0]])
    assert oracle.sample_rows(p, torch.tensor([0.0, 0.3])).tolist() == [0, 2]
    assert oracle.sample_rows(p[:1].repeat(4, 1), torch.tensor([0.24, 0.25, 0.74, 0.99])).tolist() == [0, 2, 2, 3]


def test_native_registration_shim_exports_the_reference_hook():
    """csrc/pybind/register_ops.cc: the module must export rtp_llm::registerPyModuleOps(pybind11::module&) -- the one
    symbol the reference links per build flavour (bindings/RegisterOps.h:9) -- and expose the op classes / functions."""
    import subprocess
    from rtp_llm_amd import native_ops
    from rtp_llm_amd.build import build_pybind
    path = build_pybind(verbose=False)
    syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    assert "_ZN7rtp_llm19registerPyModuleOpsERN8pybind117module_E" in syms
    ops = native_ops.load()
    for name in ("Mi355RopeKVCacheDecodeOp", "Mi355PagedAttnDecodeOp", "Mi355WeightOnlyLinear", "Mi355AttnParams", "AttentionConfigs",
                 "LayerKVCache", "PyAttentionInputs", "rmsnorm", "fused_add_rmsnorm", "silu_and_mul", "embedding", "greedy_argmax"):
        assert hasattr(ops, name), name
    for cls in (ops.Mi355RopeKVCacheDecodeOp, ops.Mi355PagedAttnDecodeOp):
        assert callable(getattr(cls, "prepare")) and callable(getattr(cls, "forward"))
    assert callable(ops.Mi355AttnParams.update_kv_cache_offset) and callable(ops.Mi355AttnParams.prepare_in_place)
    with pytest.raises(RuntimeError):                     # TORCH_CHECK -> RuntimeError, and no CPU path
        ops.rmsnorm(torch.zeros(2, 8).half(), torch.zeros(2, 8).half(), torch.ones(8).half(), 1e-6)


def test_fused_ops_refuse_cpu_tensors_and_bad_shapes():
    """The fused small-batch entry points are product path: no CPU fallback -- a CPU tensor is an error, not a slow path."""
    import pytest
    from rtp_llm_amd import _C, ops
    x = torch.zeros(2, 512, dtype=torch.float16)
    class _W:   # never reaches the library: the tensor checks come first
        K, N = 512, 256
    with pytest.raises(_C.Mi355Error):
        ops.linear_residual(x, _W(), torch.zeros(2, 256, dtype=torch.float16))
    with pytest.raises(_C.Mi355Error):
        ops.norm_linear(x, (torch.zeros(2, 32), torch.ones(512, dtype=torch.float16), 1e-6), _W())
    with pytest.raises(_C.Mi355Error):
        ops.qkv_rope_kv_write(x, _W(), None, torch.zeros(8, 32, 2), torch.zeros(2, dtype=torch.int32), torch.zeros(2, 4, dtype=torch.int32),
                              torch.zeros(1), None, 2, 1, 64, 16)


def test_sampler_and_collective_entry_points_validate_on_the_host():
    """The round-2 additions to the ABI: argument errors come back as a status before anything is launched; empty batches are a no-op."""
    l = _C.lib()
    assert l.mi355_apply_penalties(None, 0, 10, 10, None, None, None, None, None, None, 0, 0, None, None) == 0          # empty batch
    assert l.mi355_apply_penalties(None, 2, 10, 10, None, None, None, None, None, None, 0, 0, None, None) == _C.ERR_ARG
    assert l.mi355_apply_penalties(1, 2, 10, 8, None, None, None, None, None, None, 0, 0, None, None) == _C.ERR_ARG       # ld < V
    assert l.mi355_apply_penalties(1, 2, 10, 10, None, 1, None, None, None, None, 0, 4, None, None) == _C.ERR_ARG         # penalties without a history
    assert b"output_ids" in l.mi355_last_error()
    assert l.mi355_top_k_top_p_sample(None, 0, 10, 10, None, None, None, None, None, 0, None) == 0
    assert l.mi355_top_k_top_p_sample(1, 2, 10, 10, None, None, None, 1, None, 0, None) == _C.ERR_ARG                      # no uniforms
    assert l.mi355_top_k_top_p_sample(1, 2, 10, 10, None, None, 1, 1, 1, 8, None) == _C.ERR_ARG                            # ld_out < V
    assert l.mi355_ban_repeat_ngram(1, 2, 10, 10, None, 4, 1, 1, None) == _C.ERR_ARG
    assert l.mi355_ban_repeat_ngram(None, 0, 10, 10, None, 4, None, None, None) == 0
    assert l.mi355_allgather_hidden(None, 1, 1, 4, 64, None) == _C.ERR_ARG and b"not opened" in l.mi355_last_error()
    assert l.mi355_decoder_set_embedding_split(None, 1) == _C.ERR_ARG
    # activation dtype codes are validated before anything is launched
    assert l.mi355_rmsnorm_dt(1, 1, 1e-6, 2, 64, 1, 7, None) == _C.ERR_ARG and b"act_dtype" in l.mi355_last_error()
    assert l.mi355_silu_mul_dt(1, 2, 64, 1, 5, None) == _C.ERR_ARG
    assert l.mi355_allreduce_sum_dt(None, 1, 1, 2, 64, _C.ACT_BF16, None) == _C.ERR_ARG and b"not opened" in l.mi355_last_error()
    wbad = _C.Weight(1, 1, 4, 256, 64, 256, 64, 128, 9)
    assert l.mi355_linear_partial(1, 4, ctypes.byref(wbad), 1, 4, None) == _C.ERR_ARG and b"act_dtype" in l.mi355_last_error()
    assert l.mi355_decoder_attach_collective(None, None, 0) == _C.ERR_ARG
    assert l.mi355_rccl_unique_id_bytes() == 128
    assert l.mi355_rccl_unique_id(b"/nonexistent/librccl.so", ctypes.create_string_buffer(128)) < 0 and b"dlopen" in l.mi355_last_error()
    assert not l.mi355_rccl_open(b"", ctypes.create_string_buffer(128), 2, 2)                                               # rank out of range
    assert l.mi355_rccl_collective(None, None) == _C.ERR_ARG
    from rtp_llm_amd import model
    emb = torch.arange(6 * 32, dtype=torch.float16).reshape(6, 32)
    parts = [model.split_embedding_tp(emb, 2, r) for r in range(2)]
    assert torch.equal(torch.cat(parts, 1), emb) and all(p.is_contiguous() and p.shape == (6, 16) for p in parts)
    with pytest.raises(ValueError):
        model.split_embedding_tp(emb, 3, 0)


def test_rope_styles_oracle_pinned_and_product_table(golden_dir=os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")):
    """Scaled RoPE styles: the oracle reproduces the committed golden vectors bit for bit -- yarn: the reference's own torch
    restatement (mla_attention_ref.py:58-160, run by oracle/gen_rope_golden.py); llama3: transformers' published formula the
    reference's Llama3Rope implements -- and the table the product builds equals the oracle's; HF config mapping as models/llama.py."""
    import numpy as np
    from rtp_llm_amd import loader
    g = np.load(os.path.join(golden_dir, "rope_styles.npz"))
    for tag in ("yarn_a", "yarn_b"):
        dim, base, factor, orig, npos = g[tag + "_cfg"]
        sc = {"type": "yarn", "factor": float(factor), "original_max_position_embeddings": int(orig), "beta_fast": 32, "beta_slow": 1}
        inv, ms = oracle.rope_inv_freq(int(dim), float(base), sc)
        assert torch.equal(inv, torch.tensor(g[tag + "_inv_freq"])) and abs(ms - float(g[tag + "_mscale"][0])) < 1e-12
        t = oracle.rope_cos_sin_scaled(int(dim), float(base), int(npos), sc)
        assert torch.equal(t[..., 0], torch.tensor(g[tag + "_cos"])) and torch.equal(t[..., 1], torch.tensor(g[tag + "_sin"]))
        cfg = model.ModelConfig("t", 1, 256, 2, 2, int(dim), 512, 1024, rope_theta=float(base), max_pos=int(npos), rope_scaling=sc)
        assert torch.equal(model.rope_table(cfg, "cpu"), t)
    for tag in ("llama3_a", "llama3_b"):
        hd, theta, f, lo, hi, old = g[tag + "_cfg"]
        sc = {"rope_type": "llama3", "factor": float(f), "low_freq_factor": float(lo), "high_freq_factor": float(hi),
              "original_max_position_embeddings": int(old)}
        inv, ms = oracle.rope_inv_freq(int(hd), float(theta), sc)
        assert torch.equal(inv, torch.tensor(g[tag + "_inv_freq"])) and ms == 1.0
        cfg = model.ModelConfig("t", 1, 256, 2, 2, int(hd), 512, 1024, rope_theta=float(theta), max_pos=128, rope_scaling=sc)
        assert torch.equal(model.rope_table(cfg, "cpu"), oracle.rope_cos_sin_scaled(int(hd), float(theta), 128, sc))
    # linear == the reference's genBaseCache with rope_scale (RopeCache.cc:16-41): positions / scale
    lin = model.ModelConfig("t", 1, 256, 2, 2, 64, 512, 1024, rope_theta=1e4, max_pos=64, rope_scaling={"type": "linear", "factor": 4.0})
    assert torch.allclose(model.rope_table(lin, "cpu"), oracle.rope_cos_sin(64, 1e4, 16, 4.0), atol=1e-6)
    hf = {"num_attention_heads": 4, "num_hidden_layers": 1, "hidden_size": 256, "intermediate_size": 512, "vocab_size": 1024,
          "rope_theta": 500000.0, "max_position_embeddings": 131072,
          "rope_scaling": {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}}
    mc, _ = loader.config_from_hf(hf)
    assert mc.rope_scaling["rope_type"] == "llama3"
    # dynamic-NTK styles (rotary_position_embedding.h:889-902, :925-951): for DECODE the base is a function of the position (the writer
    # passes the cached length as seq_len) -- one base per table row.  Host table == oracle table bit for bit (two independent
    # restatements), rows inside the original context carry the unchanged base, the bases match the header's formulas by hand.
    mc, _ = loader.config_from_hf({**hf, "rope_scaling": {"type": "dynamic", "factor": 2.0}})
    assert mc.rope_scaling == {"type": "dynamic", "factor": 2.0, "original_max_position_embeddings": 131072}     # gpt_neox.py:128-134
    mc, _ = loader.config_from_hf({**{k: v for k, v in hf.items() if k != "rope_scaling"}, "use_dynamic_ntk": True, "seq_length": 2048})
    assert mc.rope_scaling == {"rope_type": "qwen_dynamic", "original_max_position_embeddings": 2048}            # models/qwen.py:293-295
    with pytest.raises(NotImplementedError):
        loader.config_from_hf({**hf, "rope_scaling": {"type": "longrope", "factor": 2.0}})
    for rs in ({"rope_type": "dynamic", "factor": 4.0, "original_max_position_embeddings": 64}, {"rope_type": "qwen_dynamic", "original_max_position_embeddings": 64}):
        cfg = model.ModelConfig("t", 1, 256, 4, 2, 64, 512, 1024, rope_theta=1e4, max_pos=300, rope_scaling=rs)
        tab = model.rope_table(cfg, "cpu")
        assert torch.equal(tab, oracle.rope_cos_sin_scaled(64, 1e4, 300, rs))
        chan = torch.arange(0, 64, 2).float() / 64
        plain = torch.arange(300).float()[:, None] / torch.pow(torch.tensor(1e4), chan)[None, :]            # rope_inv_freq (:324-327): t / base^(2i/d)
        assert torch.equal(tab[:65, :, 0], plain[:65].cos()) and float((tab[65:, :, 0] - plain[65:].cos()).abs().max()) > 0.5
        bases = oracle.rope_dynamic_ntk_bases(64, 1e4, 300, rs)
        if rs["rope_type"] == "dynamic":
            assert abs(float(bases[128]) - 1e4 * (4.0 * 128 / 64 - 3.0) ** (64 / 62.0)) < 1e-2 * 1e4 * 1e-3
        else:   # 128 = 2 x 64: log2 + 1 = 2 -> 2^2 - 1 = 3; 129: ceil(2.01) = 3 -> 7
            assert abs(float(bases[128]) - 1e4 * 3.0 ** (64 / 62.0)) < 1.0 and abs(float(bases[129]) - 1e4 * 7.0 ** (64 / 62.0)) < 1.0
        assert float(bases[64]) == 1e4 and float(bases[65]) > 1e4
    # "dynamic" pinned by the reference's own torch form of the style (DeepseekV3DynamicNTKScalingRotaryEmbedding,
    # rtp_llm/models/rotary_embedding/deepseek_rotary_embedding.py:80-113, executed by oracle/gen_rope_golden.py): the inverse frequencies
    # it rebuilds for a length S are those of the oracle's / the product's base at S (the class raises the base in double precision,
    # the device code and its restatements in fp32: 2e-6 relative), inside the original context the plain base; and the table row the
    # class holds for its last position S - 1 equals that position rotated with the base of S
    for tag in ("dynntk_a", "dynntk_b"):
        dim, base, factor, orig = (float(v) for v in g[tag + "_cfg"])
        dim, orig = int(dim), int(orig)
        rs = {"rope_type": "dynamic", "factor": factor, "original_max_position_embeddings": orig}
        lens = [int(v) for v in g[tag + "_lens"]]
        bases_o = oracle.rope_dynamic_ntk_bases(dim, base, max(lens) + 1, rs)
        bases_p = model.dynamic_ntk_base(dim, base, torch.arange(max(lens) + 1), rs)
        assert torch.equal(bases_o, bases_p)
        chan = torch.arange(0, dim, 2).float() / dim
        for i, S in enumerate(lens):
            inv_ref = torch.from_numpy(g[tag + "_inv_freq"][i])
            inv = 1.0 / torch.pow(bases_o[S], chan)
            assert torch.allclose(inv, inv_ref, rtol=2e-6, atol=0), (tag, S, float(((inv - inv_ref) / inv_ref).abs().max()))
            if S <= orig:
                assert float(bases_o[S]) == base
            ang = (S - 1) * inv
            assert torch.allclose(ang.cos(), torch.from_numpy(g[tag + "_cos_last"][i]), atol=2e-4 * S / orig + 1e-6)
            assert torch.allclose(ang.sin(), torch.from_numpy(g[tag + "_sin_last"][i]), atol=2e-4 * S / orig + 1e-6)
    # the native shim tabulates the same styles from RopeConfig's field meanings (style 5 / 6, factor1 / factor2, max_pos, mscale)
    from rtp_llm_amd import native_ops
    ops = native_ops.load()
    c = ops.AttentionConfigs()
    c.head_num, c.kv_head_num, c.size_per_head, c.max_seq_len, c.rope_base = 4, 4, 128, 96, 1000000.0
    c.rope_style, c.rope_scale, c.rope_factor1, c.rope_factor2, c.rope_max_pos = 5, 4.0, 1.0, 32.0, 32768
    c.rope_mscale = 0.1 * math.log(4.0) + 1.0
    want = oracle.rope_cos_sin_scaled(128, 1e6, 96, {"type": "yarn", "factor": 4.0, "original_max_position_embeddings": 32768})
    assert torch.allclose(ops.rope_table(c), want, atol=2e-6, rtol=0)
    c.rope_base, c.rope_style, c.rope_scale, c.rope_factor1, c.rope_factor2, c.rope_max_pos, c.rope_mscale = 500000.0, 6, 8.0, 1.0, 4.0, 8192, 1.0
    want = oracle.rope_cos_sin_scaled(128, 5e5, 96, {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                                     "original_max_position_embeddings": 8192})
    assert torch.allclose(ops.rope_table(c), want, atol=2e-6, rtol=0)
    c.rope_base, c.rope_style, c.rope_scale, c.rope_max_pos, c.max_seq_len = 10000.0, 3, 4.0, 32, 96          # DynamicNTK: base(p) past 32 positions
    want = oracle.rope_cos_sin_scaled(128, 1e4, 96, {"rope_type": "dynamic", "factor": 4.0, "original_max_position_embeddings": 32})
    assert torch.allclose(ops.rope_table(c), want, atol=2e-6, rtol=0) and float((want[40] - oracle.rope_cos_sin(128, 1e4, 96)[40]).abs().max()) > 0.5
    c.rope_style = 4                                                                                            # QwenDynamicNTK
    want = oracle.rope_cos_sin_scaled(128, 1e4, 96, {"rope_type": "qwen_dynamic", "original_max_position_embeddings": 32})
    assert torch.allclose(ops.rope_table(c), want, atol=2e-6, rtol=0)
    c.rope_style = 2                                                                                            # Glm2: not on this path
    with pytest.raises(RuntimeError):
        ops.rope_table(c)


def test_host_side_plans_of_the_image_launches():
    """Host-only arithmetic of the 5-64-row launches (no kernel is launched, nothing is dereferenced): image sizes, the K-quarter plan of
    down_proj, which linears the one-launch wide GEMM takes, the deferred norm's exponent."""
    import ctypes as C
    lib = _C.lib()
    assert lib.mi355_act_image_bytes(64, 3584) == 64 * 3584 * 2 and lib.mi355_act_image_bytes(5, 3584) == 16 * 3584 * 2      # whole 16-row blocks
    assert lib.mi355_act_image_bytes(0, 3584) == 0 and lib.mi355_act_image_bytes(17, 64) == 32 * 64 * 2
    plan = lib.mi355_gemm_splitk64_plan
    plan.restype, plan.argtypes = C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_int)]
    cps = C.c_int(0)
    # Qwen2-7B down_proj: N = 3584 (224 tiles -> 56 column groups), K = 18944 (148 chunks): 4 K quarters of 37 chunks, 224 blocks
    assert plan(64, 224, 148, 4, 128, 16, C.byref(cps)) == 4 and cps.value == 37
    assert plan(8, 224, 148, 4, 128, 16, C.byref(cps)) == 4                       # the same plan at every row count
    assert plan(64, 2368, 28, 4, 128, 16, C.byref(cps)) < 0                       # gate_up: N alone fills the chip -> the wide kernel's shape
    assert plan(64, 224, 148, 8, 0, 16, C.byref(cps)) == 4 and cps.value == 37    # per-channel W8 (round 5): the same K quarters
    assert plan(64, 224, 148, 8, 128, 16, C.byref(cps)) < 0 and plan(65, 224, 148, 4, 128, 16, C.byref(cps)) < 0   # group-wise W8 / > 64 rows: not this kernel
    qok = lib.mi355_fullk64_qkv_ok
    qok.restype, qok.argtypes = C.c_int, [C.POINTER(_C.Weight), C.c_int]
    mkw = lambda K, N, wbits=4, gs=128: _C.Weight(1 << 20, 1 << 21, wbits, K, N, K, N, gs, _C.ACT_F16)
    # QKV + RoPE as one launch on an image: Qwen2-7B (K 3584, W4 or W8); hidden 8192 (64 chunks) only as a TP shard whose tile pairs leave
    # half the chip free -- Llama-3-70B tp 8: (8 + 2) heads of 128 = 1280 columns, 40 pairs; the unsplit 10240 columns: no
    assert qok(C.byref(mkw(3584, 4608)), 128) == 1 and qok(C.byref(mkw(3584, 4608, 8, 0)), 128) == 1
    assert qok(C.byref(mkw(8192, 1280)), 128) == 1 and qok(C.byref(mkw(8192, 10240)), 128) == 0 and qok(C.byref(mkw(8192, 1280, 8, 0)), 128) == 0
    ok = lib.mi355_gemm_wide_direct_ok
    ok.restype, ok.argtypes = C.c_int, [C.POINTER(_C.Weight)]
    mk = lambda K, N, wbits=4, gs=128: _C.Weight(1 << 20, 1 << 21, wbits, K, N, K, N, gs, _C.ACT_F16)   # pointers are never read here
    assert ok(C.byref(mk(3584, 37888))) == 1 and ok(C.byref(mk(3584, 4608))) == 0 and ok(C.byref(mk(3584, 37888, 8, 0))) == 1 and ok(C.byref(mk(3584, 37888, 8, 128))) == 0
    from rtp_llm_amd import ops as host_ops
    assert [host_ops.norm_exponent(torch.tensor([v])) for v in (0.5, 1.0, 1.5, 2.0, 3.9, 1e9)] == [0, 0, 1, 1, 2, 14]


def test_committed_bench_line_and_traffic_file_follow_the_contract():
    """profiles/r06_bench_default.json is the line `python bench.py` printed on an MI355X: the fields the driver and the judge read are
    there, and the static PMC traffic figure is only quoted for the kernel sources it was measured on (bench.gemm_sources_sha)."""
    import json, os, importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.load(open(os.path.join(root, "profiles", "r06_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None and "workload" in line["config"]
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / rf["avg_launch_us"] / 1e3) / rf["achieved"] < 1e-2     # algorithmic bytes / measured launch time
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert abs(line["value"] - line["config"]["batch"] / line["ms_per_step"] * 1e3) / line["value"] < 1e-3
    bench = importlib.import_module("bench")
    tj = json.load(open(os.path.join(root, "profiles", "r06_traffic.json")))["qwen2-7b-w4a16"]
    assert tj["gemm_sources_sha_over"] == list(bench.TRAFFIC_KERNEL_SOURCES)
    if tj["gemm_sources_sha"] == bench.gemm_sources_sha():          # else bench.py reports traffic: null ("stale")
        assert rf["traffic"] == int(tj["gemm_quant_bytes_per_launch"]) and rf["traffic"] >= 0.9 * rf["bytes_per_launch"]
    # round 6: the same bytes over the kernels' own durations (rocprofv3 --kernel-trace pass of tools/engine_traffic.sh), beside the eager-event fraction
    assert abs(rf["avg_kernel_us"] - tj["gemm_quant_kernel_us_per_launch"]) < 1e-2 and rf["avg_kernel_us"] < rf["avg_launch_us"]
    assert abs(rf["frac_kernel_time"] - rf["bytes_per_launch"] / rf["avg_kernel_us"] / 1e3 / rf["peak"]) < 1e-3 and rf["frac_kernel_time"] > rf["frac"]
    # N > 1: the line carries `roofline` (the headline TP layout's shard of rank 0) and `cpu_baseline` too -- dry runs of `python bench.py --gpus N` (no
    # launcher: bench.py starts its own ranks, round 6) with every rank on ONE GPU (timing meaningless, the fields are what is checked)
    for n in (2, 8):
        dj = json.loads(open(os.path.join(root, "profiles", f"r06_dryrun_selfspawn_{n}ranks_one_gpu.json")).read().strip().splitlines()[-1])
        assert dj["n_gpus"] == n and "tp_layout" in dj and "error" not in dj["tp_layout"] and "replica_layout" in dj
        assert dj["tp_layout"]["ranks_bit_identical"] is True and dj["tp_layout"]["hand_over"].startswith("write-through")
        r2, c2 = dj["roofline"], dj["cpu_baseline"]
        assert r2["bound"] == "hbm" and r2["layout"].startswith("tp") and abs(r2["frac"] - r2["achieved"] / r2["peak"]) < 1e-3 and r2["traffic"] is None
        assert abs(r2["achieved"] - r2["bytes_per_launch"] / r2["avg_launch_us"] / 1e3) / r2["achieved"] < 1e-2
        assert c2["kind"] == "port" and c2["cores"] >= 1 and c2["value"] > 0
        assert dj["replica_layout"]["roofline"]["layout"].startswith("dp")
    pj = json.load(open(os.path.join(root, "profiles", "r06_parity_greedy_ids.json")))    # tests/conftest.py: the full-width end-to-end record
    assert pj["summary"]["rows"] >= 400 and pj["summary"]["exact"] >= pj["summary"]["safe"]
    assert all(r["max_abs_logit_err"] <= r["tol"] and (r["exact"] == r["rows"] or r["safe"] < r["rows"]) for r in pj["records"])


def test_bench_gpus_n_without_a_launcher_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the shape of the driver's one-GPU command) must become the launcher
    of N ranks instead of silently running one (VERDICT r05 missing #1): the spawn plan carries the torchrun environment contract, rank 0's
    stdout is the launcher's, and a failing rank makes the run non-zero.  Reference idiom: one process per rank,
    rtp_llm/start_server.py:597-720, modules/base/rocm/test/trt_allreduce_test.py:381-470."""
    import importlib, os, subprocess, sys
    bench = importlib.import_module("bench")
    plan = bench.spawn_plan(4, ["--gpus", "4", "--steps", "8"], 29511)
    assert len(plan) == 4
    for r, p in enumerate(plan):
        e = p["env"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == (str(r), str(r), "4", "127.0.0.1", "29511")
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "MI355_BENCH_ONE_GPU" not in e
        assert p["cmd"][0] == sys.executable and p["cmd"][1].endswith("bench.py") and p["cmd"][2:] == ["--gpus", "4", "--steps", "8"]
    assert all(p["env"]["MI355_BENCH_ONE_GPU"] == "1" for p in bench.spawn_plan(2, [], 1, one_gpu=True))
    # the launcher itself, with a stand-in child: every rank sees its environment, only rank 0 writes to stdout, rc propagates
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = tmp_path / "child.py"
    child.write_text("import os, sys\nr = int(os.environ['RANK'])\nprint('rank', r, 'of', os.environ['WORLD_SIZE'], os.environ['MASTER_PORT'] != '')\n"
                     "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r and 3 or 0)\n")
    drv = ("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.spawn_ranks(3, [], child_cmd=[sys.executable, %r], grace_s=5.0))" % (root, str(child)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
    ok = subprocess.run([sys.executable, "-c", drv], env=env, capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and ok.stdout.strip() == "rank 0 of 3 True", (ok.returncode, ok.stdout, ok.stderr[-500:])
    assert "rank 1 of 3" in ok.stderr and "rank 2 of 3" in ok.stderr
    bad = subprocess.run([sys.executable, "-c", drv], env={**env, "FAIL_RANK": "2"}, capture_output=True, text=True, timeout=120)
    assert bad.returncode == 3, (bad.returncode, bad.stderr[-500:])
