"""GPU parity of the full-K fused launches (gemm_fullk.hip) against the CPU oracle:

  * mi355_linear_residual  = oracle.linear (fp16 output tensor) + fp16 residual add, at the Qwen2-7B o / down shapes and a
    ragged toy shape, for the batch heights where the kernel shape changes (M = 1, 5, 16, 17, 32, 33, 48, 64);
  * mi355_qkv_rope_kv_write = oracle.linear + oracle.apply_rope + the paged-cache writer, at the Qwen2-7B qkv shape
    (28 + 2 x 4 heads of 128) and a head_dim 64 toy shape, single- and multi-row steps, stale positions included;
  * both agree with the composed launches they replace (same tolerance), and refuse (None) what they do not take.
"""
import pytest
import torch

from oracle import oracle
from rtp_llm_amd import _C, kvcache, model, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(atol=1e-2, rtol=1e-2)
MS = (1, 2, 4, 5, 7, 8, 9, 16, 17, 32, 33, 48, 64)   # <= 4 / <= 8 rows: one / two dense activation loads per chunk (gemm_fullk.hip XL)


@pytest.fixture(scope="module", autouse=True)
def _require_native():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _C.lib()


W8 = -8   # "group size" of the per-channel INT8 format (load-time autoquant, device_impl.py:183-222) in the parameter lists below


def _w4(K, N, seed, group_size=128, gate_up=False):
    """(packed weight, dense fp32 reference); group_size 0 = fp16 weights (the draft models of speculative decoding), W8 = per-channel INT8."""
    gen = torch.Generator(device=DEV).manual_seed(seed)
    if group_size == 0:
        c_dev = model.synth_linear(K, N, "fp16", DEV, gen)
        return c_dev.pack(), c_dev.w.float().cpu()
    if group_size == W8:
        c_dev = model.synth_linear(K, N, "int8", DEV, gen)
        return c_dev.pack(gate_up=gate_up), oracle.dequant_int8(c_dev.q.cpu(), c_dev.scales.cpu())
    c_dev = model.synth_linear(K, N, "w4", DEV, gen, group_size=group_size, zeros="centered")
    c = model.weights_to({"w": c_dev}, "cpu")["w"]
    return c_dev.pack(), oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size)


@pytest.mark.parametrize("K,N,gs", [(3584, 3584, 128), (18944, 3584, 128), (512, 272, 128), (1024, 256, 64), (512, 128, 32), (896, 896, 0), (4864, 896, 0)],
                         ids=["o", "down", "ragged-n", "g64", "g32", "fp16-o-0.5b", "fp16-down-0.5b"])
def test_linear_residual_vs_oracle(K, N, gs):
    packed, W = _w4(K, N, K + N, gs)
    x = (torch.randn(64, K, generator=torch.Generator().manual_seed(3)) * 0.5).half()
    res = (torch.randn(64, N, generator=torch.Generator().manual_seed(5)) * 2.0).half()
    bias = (torch.randn(N, generator=torch.Generator().manual_seed(4)) * 0.1).half()
    y = oracle.linear(x, W, bias)                                   # fp16 tensor, as the reference's linear returns it
    ref = (y.float() + res.float()).half()
    xd, rd, bd = x.to(DEV), res.to(DEV), bias.to(DEV)
    for M in MS:
        out = ops.linear_residual(xd[:M].contiguous(), packed, rd[:M].contiguous(), bd)
        assert out is not None, "W4 group-wise / fp16, K % 128 == 0: the fused kernel must take it"
        torch.cuda.synchronize()
        err = (out.cpu().float() - ref[:M].float()).abs().max()
        assert torch.allclose(out.cpu().float(), ref[:M].float(), **TOL), f"M={M}: max err {err}"
        comp = (ops.linear(xd[:M].contiguous(), packed, bd).float() + rd[:M].float()).half()      # the two launches it replaces
        assert torch.allclose(out.float(), comp.float(), **TOL)
    # in place on the residual stream, as the step driver calls it
    r2 = rd[:16].clone()
    ops.linear_residual(xd[:16].contiguous(), packed, r2, bd, out=r2)
    assert torch.equal(r2, ops.linear_residual(xd[:16].contiguous(), packed, rd[:16].contiguous(), bd))


def test_linear_residual_refuses_other_formats():
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.zeros(4, 512, dtype=torch.float16, device=DEV); r = torch.zeros(4, 256, dtype=torch.float16, device=DEV)
    p = model.synth_linear(512, 256, "int8", DEV, gen).pack()
    assert ops.linear_residual(x, p, r) is None
    p = model.synth_linear(512, 256, "w4", DEV, gen).pack()
    assert ops.linear_residual(torch.zeros(65, 512, dtype=torch.float16, device=DEV), p, torch.zeros(65, 256, dtype=torch.float16, device=DEV)) is None
    with pytest.raises(_C.Mi355Error):
        ops.linear_residual(x, p, torch.zeros(4, 128, dtype=torch.float16, device=DEV))


@pytest.mark.parametrize("nh,nkv,hd,hidden,page,q_len,gs", [(28, 4, 128, 3584, 16, 1, 128), (28, 4, 128, 3584, 16, 4, 128), (4, 2, 64, 512, 8, 1, 128),
                                                            (8, 1, 128, 1024, 16, 2, 128), (14, 2, 64, 896, 16, 1, 0)],
                         ids=["qwen2-7b", "qwen2-7b-rows4", "hd64", "mqa-rows2", "fp16-qwen2-0.5b"])
def test_qkv_rope_kv_write_vs_oracle(nh, nkv, hd, hidden, page, q_len, gs):
    N = (nh + 2 * nkv) * hd
    packed, W = _w4(hidden, N, hidden + N, gs)
    max_blocks, nblk = 8, 512
    cfg = model.ModelConfig("t", 1, hidden, nh, nkv, hd, 64, 128, max_pos=max_blocks * page)
    cs = oracle.rope_cos_sin(hd, cfg.rope_theta, cfg.max_pos)
    bias = (torch.randn(N, generator=torch.Generator().manual_seed(4)) * 0.1).half()
    for T in sorted({q_len, 2 * q_len, 4, 5 * q_len, 8, 16 * q_len, (64 // q_len) * q_len} - {0}):
        if T % q_len: continue
        nseq = T // q_len
        g = torch.Generator().manual_seed(T)
        x = (torch.randn(T, hidden, generator=g) * 0.5).half()
        start = torch.randint(0, max_blocks * page - q_len, (nseq,), generator=g)
        pos = (start[:, None] + torch.arange(q_len)[None, :]).reshape(-1).to(torch.int32)
        bt = torch.randperm(nblk, generator=g)[: nseq * max_blocks].reshape(nseq, max_blocks).to(torch.int32)
        kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV)
        q = ops.qkv_rope_kv_write(x.to(DEV), packed, bias.to(DEV), cs.to(DEV), pos.to(DEV), bt.to(DEV), kv, sc, nh, nkv, hd, page, q_len)
        assert q is not None
        torch.cuda.synchronize()
        qkv = oracle.linear(x, W, bias)
        qh = qkv[:, : nh * hd].reshape(T, nh, hd)
        kh = qkv[:, nh * hd: (nh + nkv) * hd].reshape(T, nkv, hd)
        vh = qkv[:, (nh + nkv) * hd:].reshape(T, nkv, hd)
        q_ref, k_ref = oracle.apply_rope(qh, pos, cs), oracle.apply_rope(kh, pos, cs)
        assert torch.allclose(q.cpu().float(), q_ref.float(), **TOL), f"T={T}: q max err {(q.cpu().float() - q_ref.float()).abs().max()}"
        for t in range(T):
            K, V, _, _ = kvcache.read_tokens(kv, sc, bt[t // q_len], int(pos[t]) + 1)
            assert torch.allclose(K[-1].cpu().float(), k_ref[t].float(), **TOL)
            assert torch.allclose(V[-1].cpu().float(), vh[t].float(), **TOL)
        # the composed launches it replaces: same numbers within the same tolerance, same cache image
        kv2, sc2 = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV)
        y = ops.linear(x.to(DEV), packed, None)
        q2 = ops.rope_kv_write_rows(y, bias.to(DEV), cs.to(DEV), pos.to(DEV), bt.to(DEV), kv2, sc2, nh, nkv, hd, page, q_len)
        assert torch.allclose(q.float(), q2.float(), **TOL) and torch.allclose(kv.float(), kv2.float(), **TOL)


def test_qkv_rope_kv_write_stale_rows_and_int8_refusal():
    nh, nkv, hd, hidden, page = 4, 2, 64, 512, 8
    N = (nh + 2 * nkv) * hd
    packed, _ = _w4(hidden, N, 9)
    max_blocks, nblk = 4, 32
    cfg = model.ModelConfig("t", 1, hidden, nh, nkv, hd, 64, 128, max_pos=max_blocks * page)
    cs = oracle.rope_cos_sin(hd, cfg.rope_theta, cfg.max_pos).to(DEV)
    x = (torch.randn(4, hidden, generator=torch.Generator().manual_seed(1)) * 0.5).half().to(DEV)
    pos = torch.tensor([3, -1, max_blocks * page + 7, 5], dtype=torch.int32, device=DEV)      # ok, padding row, past the table, ok
    bt = torch.arange(4 * max_blocks, dtype=torch.int32, device=DEV).reshape(4, max_blocks)
    bt[3, 0] = nblk + 5                                                                      # stale block id
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV)
    oob = torch.zeros(1, dtype=torch.int32, device=DEV)
    before = kv.clone()
    q = ops.qkv_rope_kv_write(x, packed, None, cs, pos, bt, kv, sc, nh, nkv, hd, page, 1, oob)
    torch.cuda.synchronize()
    assert q is not None and int(oob.item()) == 2            # rows 2 and 3 refused, row 1 is padding (not an error)
    changed = (kv != before).reshape(nblk, -1).any(dim=1).nonzero().flatten().tolist()
    assert changed == [0]                                     # only row 0's page was written
    kv8, sc8 = kvcache.alloc_layer_cache(nblk, nkv, page, hd, True, DEV)
    assert ops.qkv_rope_kv_write(x, packed, None, cs, pos, bt, kv8, sc8, nh, nkv, hd, page, 1) is None


@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 8, 9, 16])
def test_fused_norm_chain_vs_oracle(M):
    """o_proj + residual (leaving per-tile sums of squares) -> RMSNorm on load + gate_up + SiLU-gate, and -> RMSNorm on load +
    QKV + RoPE + KV write: the launches of a small-batch layer without norm kernels, against oracle.rmsnorm + oracle.linear."""
    cfg = model.QWEN2_7B
    H, I, nh, nkv, hd, page = cfg.hidden, 9472, cfg.nh, cfg.nkv, cfg.hd, 16      # 2 I / 16 = 1184 tiles: the 5-tiles-per-block shape, ragged last block
    wo, Wo = _w4(H, H, 1)
    gen = torch.Generator(device=DEV).manual_seed(2)
    cg = model.synth_linear(H, 2 * I, "w4", DEV, gen, zeros="centered")
    wg = cg.pack(gate_up=True)
    cgc = model.weights_to({"w": cg}, "cpu")["w"]
    Wg = oracle.dequant_groupwise(cgc.q, cgc.z_eff, cgc.scales, cgc.group_size)
    wq, Wq = _w4(H, (nh + 2 * nkv) * hd, 3)
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, H, generator=g) * 0.5).half()
    res = (torch.randn(M, H, generator=g) * 3.0).half()
    gamma = (1.0 + 0.2 * torch.randn(H, generator=g)).half()
    eps = 1e-6
    # ---- producer
    ssq = torch.zeros(16, H // 16, dtype=torch.float32, device=DEV)
    h = ops.linear_residual(x.to(DEV), wo, res.to(DEV), tile_sumsq=ssq)
    torch.cuda.synchronize()
    h_ref = (oracle.linear(x, Wo, None).float() + res.float()).half()
    assert torch.allclose(h.cpu().float(), h_ref.float(), **TOL)
    assert torch.allclose(ssq[:M].sum(1).cpu(), (h.cpu().float() ** 2).sum(1), rtol=1e-5)      # exact partial sums of what was stored
    # ---- consumers: normalise the rows the producer stored
    xn_ref = oracle.rmsnorm(h.cpu(), gamma, eps)
    norm = (ssq, gamma.to(DEV), eps)
    act = ops.norm_linear(h, norm, wg, None, _C.EPI_SILU_MUL)
    assert act is not None
    act_ref = oracle.silu_mul(oracle.linear(xn_ref, Wg, None))
    assert torch.allclose(act.cpu().float(), act_ref.float(), **TOL), (act.cpu().float() - act_ref.float()).abs().max()
    max_blocks, nblk = 8, 128
    c2 = model.ModelConfig("t", 1, H, nh, nkv, hd, 64, 128, max_pos=max_blocks * page)
    cs = oracle.rope_cos_sin(hd, c2.rope_theta, c2.max_pos)
    pos = torch.randint(0, max_blocks * page, (M,), generator=g).to(torch.int32)
    bt = torch.randperm(nblk, generator=g)[: M * max_blocks].reshape(M, max_blocks).to(torch.int32)
    bias = (torch.randn((nh + 2 * nkv) * hd, generator=g) * 0.1).half()
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV)
    q = ops.qkv_rope_kv_write(h, wq, bias.to(DEV), cs.to(DEV), pos.to(DEV), bt.to(DEV), kv, sc, nh, nkv, hd, page, 1, None, norm)
    assert q is not None
    torch.cuda.synchronize()
    qkv = oracle.linear(xn_ref, Wq, bias)
    q_ref = oracle.apply_rope(qkv[:, : nh * hd].reshape(M, nh, hd), pos, cs)
    k_ref = oracle.apply_rope(qkv[:, nh * hd: (nh + nkv) * hd].reshape(M, nkv, hd), pos, cs)
    assert torch.allclose(q.cpu().float(), q_ref.float(), **TOL)
    for t in range(M):
        K, V, _, _ = kvcache.read_tokens(kv, sc, bt[t], int(pos[t]) + 1)
        assert torch.allclose(K[-1].cpu().float(), k_ref[t].float(), **TOL)
    # more than 16 rows: not taken with a fused norm
    ssq2 = torch.zeros(32, H // 16, dtype=torch.float32, device=DEV)
    assert ops.norm_linear(torch.zeros(17, H, dtype=torch.float16, device=DEV), (ssq2, gamma.to(DEV), eps), wg) is None


# ---------------------------------------------------------------------------------------------------------------- 17-64 rows
# gemm_fullk64.hip: the same two fused launches with the activations handed over as an image (include/mi355_decode.h,
# mi355_act_image_*), against the same oracle composition and tolerance as the row-major forms above.
MS64 = (1, 5, 13, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64)     # the engine takes these launches from 5 rows up; the kernels serve any 1..64


def _img_index(M, K):
    """element index of x[m][k] in the image: the formula of include/mi355_decode.h, restated in torch"""
    m = torch.arange(M)[:, None]; k = torch.arange(K)[None, :]
    mblk = (M + 15) // 16
    return ((((k // 32) * mblk + m // 16) * 64 + ((k % 32) // 8) * 16 + m % 16) * 8 + k % 8).reshape(-1)


@pytest.mark.parametrize("M,K", [(1, 64), (17, 64), (64, 3584), (33, 512), (48, 3584)])
def test_act_image_layout_and_round_trip(M, K):
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(M + K)).half().to(DEV)
    img = ops.act_image_pack(x)
    torch.cuda.synchronize()
    assert img.data.numel() == ((M + 15) // 16) * 16 * K
    assert torch.equal(img.data.cpu()[_img_index(M, K)].reshape(M, K), x.cpu())      # documented index formula
    assert torch.equal(img.unpack(), x)


@pytest.mark.parametrize("K,N,gs", [(3584, 3584, 128), (512, 272, 128), (1024, 256, 64), (512, 128, 32), (4096, 1024, 128), (5120, 512, 128),
                                    (3584, 3584, W8), (512, 272, W8), (3840, 512, W8), (8192, 512, 128)],
                         ids=["o", "ragged-n", "g64", "g32", "k4096-three-chunk-slices", "k5120", "w8-o", "w8-ragged-n", "w8-k3840", "k8192-five-chunk-slices-row-split"])
def test_linear_residual_img_vs_oracle(K, N, gs):
    packed, W = _w4(K, N, K + N, gs)
    x = (torch.randn(64, K, generator=torch.Generator().manual_seed(3)) * 0.5).half()
    res = (torch.randn(64, N, generator=torch.Generator().manual_seed(5)) * 2.0).half()
    bias = (torch.randn(N, generator=torch.Generator().manual_seed(4)) * 0.1).half()
    ref = (oracle.linear(x, W, bias).float() + res.float()).half()
    xd, rd, bd = x.to(DEV), res.to(DEV), bias.to(DEV)
    for M in MS64:
        ssq = torch.zeros(M, (N // 16 + 3) & ~3, dtype=torch.float32, device=DEV)
        out = ops.linear_residual_img(ops.act_image_pack(xd[:M].contiguous()), packed, rd[:M].contiguous(), bd, tile_sumsq=ssq)
        assert out is not None, "W4 group-wise, 17-64 rows, K <= 5760: the image kernel must take it"
        torch.cuda.synchronize()
        err = (out.cpu().float() - ref[:M].float()).abs().max()
        assert torch.allclose(out.cpu().float(), ref[:M].float(), **TOL), f"M={M}: max err {err}"
        assert torch.allclose(ssq[:, : N // 16].sum(1).cpu(), (out.cpu().float() ** 2).sum(1), rtol=1e-5)   # exact partial sums of what was stored
        if gs != W8:
            comp = ops.linear_residual(xd[:M].contiguous(), packed, rd[:M].contiguous(), bd)        # the row-major launch of the same contract
            assert torch.allclose(out.float(), comp.float(), **TOL)
    # in place on the residual stream, as the step driver calls it; no bias
    r2 = rd.clone()
    ops.linear_residual_img(ops.act_image_pack(xd), packed, r2, None, out=r2)
    assert torch.equal(r2, ops.linear_residual_img(ops.act_image_pack(xd), packed, rd, None))


def test_img_launches_refuse_other_shapes():
    gen = torch.Generator(device=DEV).manual_seed(1)
    p = model.synth_linear(512, 256, "w4", DEV, gen).pack()
    p16 = model.synth_linear(512, 256, "fp16", DEV, gen).pack()                                                                # 16-bit weights: the composed launches
    r = torch.zeros(32, 256, dtype=torch.float16, device=DEV)
    assert ops.linear_residual_img(ops.act_image_pack(torch.zeros(32, 512, dtype=torch.float16, device=DEV)), p16, r) is None
    p8 = model.synth_linear(3968, 256, "int8", DEV, gen).pack()                                                                 # W8: 31 chunks, past its two-chunk slices
    assert ops.linear_residual_img(ops.act_image_pack(torch.zeros(32, 3968, dtype=torch.float16, device=DEV)), p8, r) is None
    pk = model.synth_linear(9728, 256, "w4", DEV, gen).pack()                                                                    # 76 chunks: past the five-chunk slices
    assert ops.linear_residual_img(ops.act_image_pack(torch.zeros(32, 9728, dtype=torch.float16, device=DEV)), pk, r) is None
    pw = model.synth_linear(5888, 4608, "w4", DEV, gen).pack()                                                                   # 46 chunks x 144 tile pairs: no room for the row split
    rw = torch.zeros(64, 4608, dtype=torch.float16, device=DEV)                                                                  # that the five-chunk slices need above 32 rows
    assert ops.linear_residual_img(ops.act_image_pack(torch.zeros(64, 5888, dtype=torch.float16, device=DEV)), pw, rw) is None


@pytest.mark.parametrize("nh,nkv,hd,hidden,page,q_len,gs", [(28, 4, 128, 3584, 16, 1, 128), (28, 4, 128, 3584, 16, 4, 128), (4, 2, 64, 512, 8, 1, 128),
                                                            (8, 1, 128, 1024, 16, 2, 64), (6, 2, 64, 512, 16, 1, 32), (28, 4, 128, 3584, 16, 1, W8), (4, 2, 64, 512, 8, 2, W8)],
                         ids=["qwen2-7b", "qwen2-7b-rows4", "hd64", "mqa-rows2-g64", "hd64-g32", "qwen2-7b-w8", "hd64-rows2-w8"])
def test_qkv_rope_kv_write_img_vs_oracle(nh, nkv, hd, hidden, page, q_len, gs):
    N = (nh + 2 * nkv) * hd
    packed, W = _w4(hidden, N, hidden + N, gs)
    max_blocks, nblk = 8, 1024
    cfg = model.ModelConfig("t", 1, hidden, nh, nkv, hd, 64, 128, max_pos=max_blocks * page)
    cs = oracle.rope_cos_sin(hd, cfg.rope_theta, cfg.max_pos)
    bias = (torch.randn(N, generator=torch.Generator().manual_seed(4)) * 0.1).half()
    for T in sorted({(t // q_len) * q_len for t in (5 + q_len - 1, 16, 17 + q_len - 1, 32, 33 + q_len - 1, 48, 64)}):
        nseq = T // q_len
        g = torch.Generator().manual_seed(T)
        x = (torch.randn(T, hidden, generator=g) * 0.5).half()
        start = torch.randint(0, max_blocks * page - q_len, (nseq,), generator=g)
        pos = (start[:, None] + torch.arange(q_len)[None, :]).reshape(-1).to(torch.int32)
        if q_len > 1:
            pos[-1] = -1                                     # a padding row of a multi-row step: q produced, nothing stored
        bt = torch.randperm(nblk, generator=g)[: nseq * max_blocks].reshape(nseq, max_blocks).to(torch.int32)
        kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV)
        before = kv.clone()
        q = ops.qkv_rope_kv_write_img(ops.act_image_pack(x.to(DEV)), packed, bias.to(DEV), cs.to(DEV), pos.to(DEV), bt.to(DEV), kv, sc, nh, nkv, hd, page, q_len)
        assert q is not None
        torch.cuda.synchronize()
        qkv = oracle.linear(x, W, bias)
        qh = qkv[:, : nh * hd].reshape(T, nh, hd)
        kh = qkv[:, nh * hd: (nh + nkv) * hd].reshape(T, nkv, hd)
        vh = qkv[:, (nh + nkv) * hd:].reshape(T, nkv, hd)
        pos_c = pos.clamp(min=0)
        q_ref, k_ref = oracle.apply_rope(qh, pos_c, cs), oracle.apply_rope(kh, pos_c, cs)
        assert torch.allclose(q.cpu().float(), q_ref.float(), **TOL), f"T={T}: q max err {(q.cpu().float() - q_ref.float()).abs().max()}"
        for t in range(T):
            if pos[t] < 0: continue
            K, V, _, _ = kvcache.read_tokens(kv, sc, bt[t // q_len], int(pos[t]) + 1)
            assert torch.allclose(K[-1].cpu().float(), k_ref[t].float(), **TOL)
            assert torch.allclose(V[-1].cpu().float(), vh[t].float(), **TOL)
        # the composed launches it replaces leave the same cache image (same tolerance) and touch the same pages only
        kv2, sc2 = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV)
        y = ops.linear(x.to(DEV), packed, None)
        q2 = ops.rope_kv_write_rows(y, bias.to(DEV), cs.to(DEV), pos.to(DEV), bt.to(DEV), kv2, sc2, nh, nkv, hd, page, q_len)
        assert torch.allclose(q.float(), q2.float(), **TOL) and torch.allclose(kv.float(), kv2.float(), **TOL)
        assert torch.equal((kv != before).reshape(nblk, -1).any(1), (kv2 != before).reshape(nblk, -1).any(1))


def test_qkv_rope_kv_write_img_stale_rows_and_int8_refusal():
    nh, nkv, hd, hidden, page = 4, 2, 64, 512, 8
    N = (nh + 2 * nkv) * hd
    packed, _ = _w4(hidden, N, 9)
    max_blocks, nblk, T = 4, 256, 20
    cfg = model.ModelConfig("t", 1, hidden, nh, nkv, hd, 64, 128, max_pos=max_blocks * page)
    cs = oracle.rope_cos_sin(hd, cfg.rope_theta, cfg.max_pos).to(DEV)
    x = ops.act_image_pack((torch.randn(T, hidden, generator=torch.Generator().manual_seed(1)) * 0.5).half().to(DEV))
    pos = torch.full((T,), 3, dtype=torch.int32, device=DEV)
    pos[1] = -1; pos[2] = max_blocks * page + 7                                            # padding row, past the table
    bt = torch.arange(T * max_blocks, dtype=torch.int32, device=DEV).reshape(T, max_blocks)
    bt[3, 0] = nblk + 5                                                                      # stale block id
    kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, False, DEV)
    oob = torch.zeros(1, dtype=torch.int32, device=DEV)
    before = kv.clone()
    q = ops.qkv_rope_kv_write_img(x, packed, None, cs, pos, bt, kv, sc, nh, nkv, hd, page, 1, oob)
    torch.cuda.synchronize()
    assert q is not None and int(oob.item()) == 2            # rows 2 and 3 refused, row 1 is padding (not an error)
    changed = (kv != before).reshape(nblk, -1).any(dim=1).nonzero().flatten().tolist()
    assert changed == [t * max_blocks for t in range(T) if t not in (1, 2, 3)]
    kv8, sc8 = kvcache.alloc_layer_cache(nblk, nkv, page, hd, True, DEV)
    assert ops.qkv_rope_kv_write_img(x, packed, None, cs, pos, bt, kv8, sc8, nh, nkv, hd, page, 1) is None


@pytest.mark.parametrize("M", [5, 17, 40, 64])
def test_image_producers_equal_the_row_major_launches(M):
    """RMSNorm (+ residual add) and paged attention writing images: bit for bit the row-major results at the image's addresses."""
    H = 3584
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, H, generator=g).half().to(DEV); res = torch.randn(M, H, generator=g).half().to(DEV)
    gamma = (1.0 + 0.2 * torch.randn(H, generator=g)).half().to(DEV)
    y, r = ops.add_rmsnorm(x, res, gamma, 1e-6)
    yi, ri = ops.add_rmsnorm_img(x, res, gamma, 1e-6)
    assert torch.equal(yi.unpack(), y) and torch.equal(ri, r)
    yi, _ = ops.add_rmsnorm_img(x, None, gamma, 1e-6)
    assert torch.equal(yi.unpack(), ops.rmsnorm(x, gamma, 1e-6))
    nh, nkv, hd, page, max_blocks, nblk = 28, 4, 128, 16, 8, 1024
    for int8, q_len in ((False, 1), (True, 1), (False, 4)):
        T = (M // q_len) * q_len
        nseq = T // q_len
        kv, sc = kvcache.alloc_layer_cache(nblk, nkv, page, hd, int8, DEV)
        if int8:
            kv.copy_(torch.randint(0, 256, kv.shape, generator=g, dtype=torch.int32).to(kv.dtype)); sc.copy_(torch.rand(sc.shape, generator=g) * 0.02)
        else:
            kv.copy_(torch.randn(kv.shape, generator=g).half())
        q = torch.randn(T, nh, hd, generator=g).half().to(DEV)
        start = torch.randint(0, max_blocks * page - q_len, (nseq,), generator=g)
        pos = (start[:, None] + torch.arange(q_len)[None, :]).reshape(-1).to(torch.int32).to(DEV)
        bt = torch.randperm(nblk, generator=g)[: nseq * max_blocks].reshape(nseq, max_blocks).to(torch.int32).to(DEV)
        a = ops.paged_attention_rows(q, kv, sc, bt, pos, nkv, page, q_len, max_blocks * page)
        ai = ops.paged_attention_rows_img(q, kv, sc, bt, pos, nkv, page, q_len, max_blocks * page)
        assert torch.equal(ai.unpack(), a)


@pytest.mark.parametrize("M,gmax,hscale", [(5, 1.2, 3.0), (16, 1.2, 3.0), (17, 1.2, 3.0), (32, 1.2, 3.0), (48, 30.0, 3.0), (64, 1.2, 3.0), (64, 30.0, 2000.0)],
                         ids=["5", "16", "17", "32", "48-large-gamma", "64", "64-large-gamma-massive-residual"])
def test_deferred_norm_chain_vs_oracle(M, gmax, hscale):
    """o_proj + residual (leaving gamma 2^-e h' as an image and the per-tile sums of h'^2) -> gate_up + SiLU-gate with the RMSNorm
    finished on the accumulators: the 17-64-row layer without its post-attention norm launch, against oracle.rmsnorm + oracle.linear.
    The last case puts a norm weight of 30 on a residual stream of +-2000-8000: gamma h' would overflow fp16, gamma 2^-5 h' does not."""
    cfg = model.QWEN2_7B
    H, I = cfg.hidden, cfg.inter
    wo, Wo = _w4(H, H, 1)
    gen = torch.Generator(device=DEV).manual_seed(2)
    cg = model.synth_linear(H, 2 * I, "w4", DEV, gen, zeros="centered")
    wg = cg.pack(gate_up=True)
    cgc = model.weights_to({"w": cg}, "cpu")["w"]
    Wg = oracle.dequant_groupwise(cgc.q, cgc.z_eff, cgc.scales, cgc.group_size)
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, H, generator=g) * 0.5).half()
    res = (torch.randn(M, H, generator=g) * hscale).half()
    gamma = (1.0 + 0.2 * torch.randn(H, generator=g)).half()
    gamma[::97] = gmax                                   # a few large norm weights
    eps = 1e-6
    r = ops.linear_residual_prenorm_img(ops.act_image_pack(x.to(DEV)), wo, res.to(DEV), gamma.to(DEV))
    assert r is not None
    h, xg, ssq, e = r
    torch.cuda.synchronize()
    assert 2.0 ** e >= gmax and (e == 0 or 2.0 ** (e - 1) < max(gmax, float(gamma.float().abs().max())))
    h_ref = (oracle.linear(x, Wo, None).float() + res.float()).half()
    assert torch.allclose(h.cpu().float(), h_ref.float(), **TOL)
    assert torch.allclose(ssq[:, : H // 16].sum(1).cpu(), (h.cpu().float() ** 2).sum(1), rtol=1e-5)
    xg_ref = (gamma.float() * 2.0 ** -e * h.cpu().float()).half()                   # one rounding from fp32
    assert torch.equal(xg.unpack().cpu(), xg_ref) and torch.isfinite(xg_ref.float()).all()
    act = ops.linear_deferred_norm_img(xg, (ssq, eps, e), wg, None, _C.EPI_SILU_MUL)
    assert act is not None
    act_ref = oracle.silu_mul(oracle.linear(oracle.rmsnorm(h.cpu(), gamma, eps), Wg, None))
    err = (act.cpu().float() - act_ref.float()).abs().max()
    assert torch.allclose(act.cpu().float(), act_ref.float(), **TOL), err
    # the plain linear on an image (no deferred norm) equals the row-major wide GEMM bit for bit: same instruction stream, other addresses
    y_img = ops.linear_deferred_norm_img(ops.act_image_pack(h), None, wg, None, _C.EPI_SILU_MUL)
    y_row = ops.linear(h, wg, None, _C.EPI_SILU_MUL)
    if M > 16:      # row-major callers reach the wide kernel above 16 rows only (the staged kernel below: other summation order)
        assert torch.equal(y_img, y_row)
    else:
        assert torch.allclose(y_img.float(), y_row.float(), **TOL)
    # a narrow N cannot fill the chip in one launch: not taken
    assert ops.linear_deferred_norm_img(ops.act_image_pack(h), None, wo) is None


@pytest.mark.parametrize("K,N,gs", [(18944, 3584, 128), (3584, 8192, 128), (4096, 3584, 64), (4096, 3584, 32), (3712, 8192, 128), (18944, 3584, W8), (4096, 3584, W8)],
                         ids=["7b-down", "70b-tp8-down", "g64", "g32", "72b-tp8-down-29-chunks", "w8-7b-down", "w8-k4096"])
def test_linear_partial_img_vs_oracle(K, N, gs):
    """gemm_splitk64.hip: the slabs of a deep-K linear from an activation image sum to oracle.linear (fp32 accumulation, so the sum
    is compared before any fp16 rounding) and fold into the same residual + RMSNorm as the staged kernel's slabs."""
    packed, W = _w4(K, N, K + N, gs)
    x = (torch.randn(64, K, generator=torch.Generator().manual_seed(3)) * 0.5).half()
    ref = x.float() @ W                                      # fp32 GEMM of the dequantised weights
    for M in (1, 5, 16, 17, 32, 33, 48, 64):
        slabs = ops.linear_partial_img(ops.act_image_pack(x[:M].contiguous().to(DEV)), packed)
        assert slabs is not None and 2 <= slabs.shape[0] <= 16
        torch.cuda.synchronize()
        y = slabs.sum(0)[:, :N].cpu()
        err = (y - ref[:M]).abs().max()
        assert torch.allclose(y, ref[:M], atol=1e-2, rtol=1e-2), f"M={M}: max err {err}"
        y16 = ops.linear(x[:M].contiguous().to(DEV), packed, None)               # the composed path, rounded once to fp16
        assert torch.allclose(y.half().float(), y16.cpu().float(), atol=1e-2, rtol=1e-2)


def test_linear_partial_img_refuses_shapes_outside_its_plan():
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = ops.act_image_pack(torch.zeros(32, 512, dtype=torch.float16, device=DEV))
    assert ops.linear_partial_img(x, model.synth_linear(512, 256, "w4", DEV, gen).pack()) is None          # 4 chunks: too short for 8 K slices
    xw = ops.act_image_pack(torch.zeros(32, 3584, dtype=torch.float16, device=DEV))
    assert ops.linear_partial_img(xw, model.synth_linear(3584, 37888, "w4", DEV, gen).pack(gate_up=True)) is None   # N alone fills the chip
    assert ops.linear_partial_img(xw, model.synth_linear(3584, 3584, "int8", DEV, gen).pack()) is None


@pytest.mark.parametrize("M,gmax", [(8, 1.2), (16, 30.0), (33, 1.2), (64, 1.2)])
def test_w8_layer_on_images_vs_oracle(M, gmax):
    """Per-channel W8 (configs[1]: load-time INT8 autoquant) on the image launches of round 5: O + residual with the deferred-norm operands
    (gemm_fullk64 W8) -> gate_up + SiLU with the RMSNorm finished on the accumulators, image out (gemm_wide W8) -> down as K quarters
    (gemm_splitk64 W8), at the Qwen2-7B widths, against the oracle's rmsnorm / linear / silu_mul on the dequantised weights."""
    cfg = model.QWEN2_7B
    H, I = cfg.hidden, cfg.inter
    wo, Wo = _w4(H, H, 11, W8)
    wg, Wg = _w4(H, 2 * I, 12, W8, gate_up=True)
    wd, Wd = _w4(I, H, 13, W8)
    Wg_cols = Wg                                          # dense reference in canonical [gate | up] column order (the pack interleaves)
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, H, generator=g) * 0.5).half()
    res = (torch.randn(M, H, generator=g) * 3.0).half()
    gamma = (1.0 + 0.2 * torch.randn(H, generator=g)).half()
    gamma[::97] = gmax
    eps = 1e-6
    r = ops.linear_residual_prenorm_img(ops.act_image_pack(x.to(DEV)), wo, res.to(DEV), gamma.to(DEV))
    assert r is not None, "per-channel W8, K = 3584: the image kernel must take it"
    h, xg, ssq, e = r
    h_ref = (oracle.linear(x, Wo, None).float() + res.float()).half()
    assert torch.allclose(h.cpu().float(), h_ref.float(), **TOL), float((h.cpu().float() - h_ref.float()).abs().max())
    act_img = ops.linear_deferred_norm_img(xg, (ssq, eps, e), wg, None, _C.EPI_SILU_MUL | _C.EPI_OUT_IMAGE)
    assert isinstance(act_img, ops.ActImage), "per-channel W8 gate_up: the wide GEMM's image entry must take it"
    act_ref = oracle.silu_mul(oracle.linear(oracle.rmsnorm(h.cpu(), gamma, eps), Wg_cols, None))
    act = act_img.unpack().cpu()
    assert torch.allclose(act.float(), act_ref.float(), **TOL), float((act.float() - act_ref.float()).abs().max())
    slabs = ops.linear_partial_img(act_img, wd)
    assert slabs is not None and slabs.shape[0] == 4
    y = slabs.sum(0)[:, :H].cpu()
    ref = act.float() @ Wd
    assert torch.allclose(y, ref, atol=1e-2, rtol=1e-2), float((y - ref).abs().max())


@pytest.mark.parametrize("M", [5, 17, 40, 64])
def test_wide_gemm_writes_the_image_the_down_launch_reads(M):
    """gate_up + SiLU-gate with the output as an image == the row-major output at the image's addresses (bit for bit), and the
    whole MLP on images (gate_up -> down slabs) matches the oracle MLP."""
    cfg = model.QWEN2_7B
    H, I = cfg.hidden, cfg.inter
    gen = torch.Generator(device=DEV).manual_seed(2)
    cg = model.synth_linear(H, 2 * I, "w4", DEV, gen, zeros="centered"); wg = cg.pack(gate_up=True)
    cd = model.synth_linear(I, H, "w4", DEV, gen, zeros="centered"); wd = cd.pack()
    x = (torch.randn(M, H, generator=torch.Generator().manual_seed(M)) * 0.5).half().to(DEV)
    xi = ops.act_image_pack(x)
    act = ops.linear_deferred_norm_img(xi, None, wg, None, _C.EPI_SILU_MUL)
    act_img = ops.linear_deferred_norm_img(xi, None, wg, None, _C.EPI_SILU_MUL | _C.EPI_OUT_IMAGE)
    assert isinstance(act_img, ops.ActImage) and torch.equal(act_img.unpack(), act)
    slabs = ops.linear_partial_img(act_img, wd)
    assert slabs is not None
    cgc, cdc = model.weights_to({"w": cg}, "cpu")["w"], model.weights_to({"w": cd}, "cpu")["w"]
    Wg = oracle.dequant_groupwise(cgc.q, cgc.z_eff, cgc.scales, cgc.group_size)
    Wd = oracle.dequant_groupwise(cdc.q, cdc.z_eff, cdc.scales, cdc.group_size)
    ref = oracle.linear(oracle.silu_mul(oracle.linear(x.cpu(), Wg, None)), Wd, None)
    y = slabs.sum(0)[:, :H].half().cpu()
    assert torch.allclose(y.float(), ref.float(), **TOL), (y.float() - ref.float()).abs().max()
