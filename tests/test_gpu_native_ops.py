"""The pybind11 registration shim (csrc/pybind/register_ops.cc) on the GPU: op classes of the reference's shape
(prepare / forward / update_kv_cache_offset) against the oracle, and bit-equal to the ctypes path (same kernels)."""
import math

import pytest
import torch

from oracle import oracle
from rtp_llm_amd import _C, kvcache, model, native_ops, ops as cops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(atol=1e-2, rtol=1e-2)


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("int8", [False, True])
def test_rope_kv_and_paged_attention_op_classes(int8):
    ops = native_ops.load()
    nh, nkv, hd, page, B, nblk, M = 28, 4, 128, 16, 5, 64, 8
    c = ops.AttentionConfigs()
    c.head_num, c.kv_head_num, c.size_per_head, c.tokens_per_block, c.max_seq_len, c.rope_base = nh, nkv, hd, page, M * page, 1e6
    rope_op, attn_op = ops.Mi355RopeKVCacheDecodeOp(c), ops.Mi355PagedAttnDecodeOp(c)
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, int8, DEV)
    lk = ops.LayerKVCache()
    lk.kv_cache_base, lk.seq_size_per_block, lk.layer_id = kv, page, 0
    if int8:
        lk.kv_scale_base = sc
    bt = torch.randperm(nblk, generator=_g(1))[: B * M].reshape(B, M).to(torch.int32)
    ctx = [0, 3, 17, 40, 100]
    cs = oracle.rope_cos_sin(hd, 1e6, M * page)
    okv = oracle.OracleKV(1, B, int8)
    for b in range(B):                                      # pre-existing context, same on both sides
        K, V = torch.randn(ctx[b], nkv, hd, generator=_g(10 + b)).half(), torch.randn(ctx[b], nkv, hd, generator=_g(20 + b)).half()
        for t in range(ctx[b]):
            okv.append(0, b, K[t], V[t])
        if ctx[b]:
            Kc, Vc, ks, vs = okv.get(0, b)
            kvcache.write_tokens(kv, sc, bt[b], 0, Kc, Vc, ks, vs)
    ai = ops.PyAttentionInputs()
    ai.is_prefill, ai.sequence_lengths, ai.kv_cache_kernel_block_id_device = False, torch.tensor(ctx, dtype=torch.int32), bt.to(DEV)
    params = rope_op.prepare(ai)
    qkv = (torch.randn(B, (nh + 2 * nkv) * hd, generator=_g(3)) * 0.5).half()
    q = rope_op.forward(qkv.to(DEV), lk, params)
    out = attn_op.forward(q, lk, params)
    torch.cuda.synchronize()
    assert rope_op.oob_count() == 0
    pos = torch.tensor(ctx, dtype=torch.int32)
    q_ref = oracle.apply_rope(qkv[:, : nh * hd].reshape(B, nh, hd), pos, cs)
    k_ref = oracle.apply_rope(qkv[:, nh * hd:(nh + nkv) * hd].reshape(B, nkv, hd), pos, cs)
    v_ref = qkv[:, (nh + nkv) * hd:].reshape(B, nkv, hd)
    assert torch.allclose(q.cpu().float(), q_ref.float(), **TOL)
    for b in range(B):
        if int8:   # the oracle attends over the codes the kernel wrote for the new token (a 1-ulp fp16 difference in the rotated K may
            #        flip one; the flips are bounded below), so the attention itself is compared at 1e-2 like the fp16 case
            Kc, Vc, ksc, vsc = kvcache.read_tokens(kv, sc, bt[b], ctx[b] + 1)
            okv.forced = lambda l, bb, t, f=(Kc[-1].cpu(), ksc[-1].cpu(), Vc[-1].cpu(), vsc[-1].cpu()): f
        okv.append(0, b, k_ref[b], v_ref[b])
        K, V, ks, vs = okv.get(0, b)
        ref = oracle.attention_decode(q_ref[b], K, V, 1 / math.sqrt(hd), ks, vs).reshape(-1)
        assert torch.allclose(out[b].cpu().float(), ref.float(), atol=1e-2, rtol=1e-2), b
    if int8:
        assert okv.max_delta <= 1 and okv.flips <= 0.10 * okv.codes, (okv.flips, okv.codes)
        assert okv.max_scale_rel <= 2e-3, okv.max_scale_rel      # the scale plane the kernel wrote, against the oracle's own scales (1 fp16 ulp of the row's amax)
    # same kernels as the ctypes path: bit-equal
    q2 = cops.rope_kv_write(qkv.to(DEV), None, cs.to(DEV), pos.to(DEV), bt.to(DEV), kv, sc, nh, nkv, hd, page)
    out2 = cops.paged_decode_attention(q2, kv, sc, bt.to(DEV), (pos + 1).to(DEV), nkv, page, M * page)
    assert torch.equal(q2, q) and torch.equal(out2, out)
    # graph-replay refresh: new block ids / lengths land in the address-stable params tensors
    ptr = params.block_table.data_ptr()
    params.update_kv_cache_offset(torch.flip(bt, [0]).contiguous().to(DEV))
    ai.sequence_lengths = torch.tensor([c + 1 for c in ctx], dtype=torch.int32)
    params.prepare_in_place(ai)
    assert params.block_table.data_ptr() == ptr and params.seq_lens.cpu().tolist() == [c + 2 for c in ctx]
    with pytest.raises(RuntimeError):
        ai.is_prefill = True
        rope_op.prepare(ai)


def test_weight_only_linear_and_free_functions():
    ops = native_ops.load()
    c = model.synth_linear(1024, 768, "w4", "cpu", _g(4))
    p = c.pack().to(DEV)
    bias = (torch.randn(768, generator=_g(5)) * 0.1).half()
    lin = ops.Mi355WeightOnlyLinear(p.qweight, p.meta, p.wbits, p.K, p.N, p.K_pad, p.N_pad, p.group_size, bias.to(DEV))
    x = (torch.randn(7, 1024, generator=_g(6)) * 0.5).half()
    y = lin.forward(x.to(DEV))
    ref = oracle.linear(x, oracle.dequant_groupwise(c.q, c.z_eff, c.scales, 128), bias)
    assert torch.allclose(y.cpu().float(), ref.float(), **TOL)
    assert torch.equal(y, cops.linear(x.to(DEV), p, bias.to(DEV)))
    h, r, w = x[:, :896].contiguous(), (torch.randn(7, 896, generator=_g(7))).half(), (1 + 0.1 * torch.randn(896, generator=_g(8))).half()
    out, res = torch.empty(7, 896, dtype=torch.float16, device=DEV), torch.empty(7, 896, dtype=torch.float16, device=DEV)
    ops.fused_add_rmsnorm(out, res, h.to(DEV), r.to(DEV), w.to(DEV), 1e-6)
    assert torch.equal(res.cpu(), h + r)
    assert torch.allclose(out.cpu().float(), oracle.rmsnorm(h + r, w, 1e-6).float(), atol=5e-2, rtol=5e-2)
    ops.rmsnorm(out, h.to(DEV), w.to(DEV), 1e-6, torch.cuda.current_stream().cuda_stream)     # explicit raw stream
    assert torch.allclose(out.cpu().float(), oracle.rmsnorm(h, w, 1e-6).float(), atol=5e-2, rtol=5e-2)
    gu = (torch.randn(7, 512, generator=_g(9))).half()
    so = torch.empty(7, 256, dtype=torch.float16, device=DEV)
    ops.silu_and_mul(so, gu.to(DEV))
    assert torch.allclose(so.cpu().float(), oracle.silu_mul(gu).float(), **TOL)
    logits = torch.randn(4, 5000, generator=_g(10))
    assert torch.equal(ops.greedy_argmax(logits.to(DEV)).cpu(), oracle.greedy(logits))
    with pytest.raises(RuntimeError):
        lin.forward(x[:, :512].contiguous().to(DEV))            # K mismatch -> TORCH_CHECK


def test_native_ops_take_bf16_tensors():
    """The shim picks the activation dtype from the tensors (the reference passes fp16 or bf16 through the same ops): W4 linear,
    norms, SiLU, RoPE + KV write and paged attention in bf16 through the op classes, bit-equal to the ctypes path (same kernels),
    and mixed dtypes are a TORCH_CHECK error."""
    BF = torch.bfloat16
    ops = native_ops.load()
    c = model.synth_linear(1024, 1536, "w4", "cpu", _g(41))
    pw = c.pack().to(DEV)
    lin = ops.Mi355WeightOnlyLinear(pw.qweight, pw.meta, pw.wbits, pw.K, pw.N, pw.K_pad, pw.N_pad, pw.group_size)
    x = (torch.randn(9, 1024, generator=_g(1)) * 0.5).to(BF).to(DEV)
    y = lin.forward(x)
    assert y.dtype == BF and torch.equal(y, cops.linear(x, pw))
    assert lin.forward(x.half()).dtype == torch.float16                      # the same object serves both dtypes (W4 image is dtype-free)
    w = (1 + 0.1 * torch.randn(1024, generator=_g(2))).to(BF).to(DEV)
    out = torch.empty_like(x)
    ops.rmsnorm(out, x, w, 1e-6)
    assert torch.equal(out, cops.rmsnorm(x, w, 1e-6))
    res, out2 = torch.empty_like(x), torch.empty_like(x)
    ops.fused_add_rmsnorm(out2, res, x, x, w, 1e-6)
    y2, r2 = cops.add_rmsnorm(x, x, w, 1e-6)
    assert torch.equal(out2, y2) and torch.equal(res, r2)
    with pytest.raises(RuntimeError):
        ops.rmsnorm(out, x, w.half(), 1e-6)
    # attention op classes on a bf16 cache
    nh, nkv, hd, page, B, nblk, M = 8, 2, 128, 16, 3, 16, 4
    cfg = ops.AttentionConfigs()
    cfg.head_num, cfg.kv_head_num, cfg.size_per_head, cfg.tokens_per_block, cfg.max_seq_len, cfg.rope_base = nh, nkv, hd, page, M * page, 1e6
    rope_op, attn_op = ops.Mi355RopeKVCacheDecodeOp(cfg), ops.Mi355PagedAttnDecodeOp(cfg)
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV, dtype=BF)
    lk = ops.LayerKVCache()
    lk.kv_cache_base, lk.seq_size_per_block, lk.layer_id = kv, page, 0
    inp = ops.PyAttentionInputs()
    pos = torch.tensor([0, 5, 17], dtype=torch.int32)
    bt = torch.arange(B * M, dtype=torch.int32).reshape(B, M)
    inp.is_prefill = False
    inp.sequence_lengths, inp.kv_cache_kernel_block_id_device = pos, bt.to(DEV)
    qkv = (torch.randn(B, (nh + 2 * nkv) * hd, generator=_g(3)) * 0.5).to(BF).to(DEV)
    params = rope_op.prepare(inp)
    q = rope_op.forward(qkv, lk, params)
    kv2, _ = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV, dtype=BF)
    cs = oracle.rope_cos_sin(hd, 1e6, M * page).to(DEV)
    q2 = cops.rope_kv_write(qkv, None, cs, pos.to(DEV), bt.to(DEV), kv2, None, nh, nkv, hd, page)
    assert q.dtype == BF and torch.equal(q, q2) and torch.equal(kv, kv2)
    o = attn_op.forward(q, lk, attn_op.prepare(inp))
    o2 = cops.paged_decode_attention(q2, kv2, None, bt.to(DEV), (pos + 1).to(DEV), nkv, page, M * page)
    assert o.dtype == BF and torch.equal(o, o2)
    with pytest.raises(RuntimeError):
        rope_op.forward(qkv.half(), lk, params)                                # fp16 rows against a bf16 cache
