// Weight-only dequant GEMM for the decode regime (M <= 64 per launch), gfx950.
//
//   y[M,N] = x[M,K] @ W[K,N]    W in {int4 group-wise, int8 per-channel/group-wise, fp16}
//
// Replaces the (absent) W4A16/W8A16 strategy slot of the reference's
// LinearFactory (rtp_llm/models_py/modules/factory/linear/factory.py:106-119) and
// the fp16 hipBLASLt path (impl/rocm/f16_linear.py:100-112) for M <= 64.
//
// Design (HBM-bound: every weight byte is read exactly once, coalesced 1 KiB per
// wave-load, non-temporal):
//  * weights are the MFMA *A* operand (16 output columns x 32 k per
//    v_mfma_f32_16x16x32_f16), activations the B operand (16 batch rows), so the
//    packed int4 stream is consumed straight from VGPRs: one dwordx4 per lane =
//    one 16x128 weight tile = 4 MFMA k-steps.  No LDS round trip for weights.
//  * int4 -> fp16 in-register: (nibble | 0x6400) is fp16(1024+u); exact add of
//    -(1024+z), one v_pk_mul by the group scale.  int8 uses v_perm the same way.
//  * x (tiny, L2 resident) is shared by the 4 waves of a block through LDS in a
//    fragment-major, XOR-swizzled image (conflict-free ds_read_b128).
//  * each wave keeps D chunks of weight loads in flight in a register ring
//    (buffer loads: out-of-range offsets return 0 and cost nothing, so the tail
//    needs no branches).
//  * split-K across blocks writes fp32 slabs; the consumer kernel (RoPE/KV-write,
//    add+RMSNorm, or reduce_epilogue below) folds the reduction — deterministic order.
#include "gemm_common.h"

extern "C" int mi355_gemm_smallm(const void* gp, int wbits, int group_size, int want_partial, int max_splits,
                                 mi355_stream_t stream);
extern "C" int mi355_gemm_wide(const void* gp, int wbits, int group_size, int want_partial, int max_splits,
                               mi355_stream_t stream);
extern "C" int mi355_gemm_prefill(const void* gp, int wbits, int group_size, mi355_stream_t stream);
extern "C" int mi355_gemm_fullk(const void* gp, int wbits, int group_size, const void* norm, mi355_stream_t stream);
extern "C" int mi355_gemm_fullk_residual(const void* gp, int wbits, int group_size, const void* residual_in, void* residual_out,
                                         float* ssq_out, int ssq_ld, mi355_stream_t stream);
extern "C" int mi355_gemm_fullk_residual_img(const void* gp, int wbits, int group_size, const void* residual_in, void* residual_out,
                                             float* ssq_out, int ssq_ld, const void* norm_weight, float xg_scale, void* xg_img,
                                             mi355_stream_t stream);
extern "C" int mi355_gemm_splitk64(const void* gp, int wbits, int group_size, int max_splits, mi355_stream_t stream);
extern "C" int mi355_gemm_splitk64_direct(const void* gp, int wbits, int group_size, mi355_stream_t stream);
extern "C" int mi355_gemm_wide_img(const void* gp, int wbits, int group_size, const mi355_deferred_norm_t* dn, mi355_stream_t stream);
extern "C" int mi355_gemm_fullk_rope_img(const void* gp, int wbits, int group_size, const float* cos_sin, int32_t max_pos,
                                         const int32_t* positions, const int32_t* block_table, int32_t max_blocks_per_seq,
                                         int32_t q_len, int32_t nh, const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count,
                                         mi355_stream_t stream);
extern "C" int mi355_gemm_fullk_rope(const void* gp, int wbits, int group_size, const void* norm, const float* cos_sin, int32_t max_pos,
                                     const int32_t* positions, const int32_t* block_table, int32_t max_blocks_per_seq,
                                     int32_t q_len, int32_t nh, const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count,
                                     mi355_stream_t stream);

#ifdef MI355_TUNING
int g_tune[16] = {0};
extern unsigned long long* g_wide_stamps;   // gemm_wide.hip: device buffer set by mi355_debug_ptr
#endif

namespace {

// Block shape: NWN "n-waves" share one x tile and own NBW 16-column tiles each (BN = 16*NBW*NWN columns);
// KG "k-groups" of NWN waves split the block's chunk range between them (intra-block split-K, merged
// through LDS at the end).  Small M wants many waves in flight per CU (KG = 2, narrow BN); large M wants a
// wide BN so the x tile (4*MB KiB per chunk) is amortised over >= as many weight bytes.
//
// Zero / scale on the C side.  The MFMA consumes the biased codes (1024+u or 64+u, exact in fp16) and
// accumulates  S = sum_k (bias_k + u_k) x_k  per quantisation group in fp32 (products are exact: 11 x 11 bits).
// With X0 / X1 the fp32 sums of x over the 1024-biased / 64-biased positions of the group (computed once
// per chunk by the threads that stage x, kept in LDS):
//     sum_k (u_k - z) x_k = S - (1024+z) X0 - (64+z) X1 = [S + 960 X1] + zneg (X0 + X1),   zneg = -(1024+z)
//     y += scale * that.
// Cost: 2 v_fma_mix per output element per group instead of 2 packed ops per weight pair — 2x fewer VALU
// at M <= 16, where the dequant ALU work was the measured limiter (VALU issues 1 wave-instruction per 4 cycles).
// BF: bf16 activations (x, bias, y and, for WBITS = 16, the weights).  The W4 codes enter the bf16 MFMA as 128 + u and zero /
// scale are applied on the accumulator side at EVERY row-block count (there is no packed bf16 arithmetic for an operand-side
// dequant); the reference's bf16 linear is f16_linear.py:100-112 with a bf16 tensor.
template <int WBITS, int MB, int NBW, int GS, int D, int NWN, int KG, bool BF = false>
__global__ __launch_bounds__(64 * NWN * KG) void gemm_wq_kernel(const GemmParams p) {
    constexpr int LPC    = WBITS / 4;              // wave-loads per (tile, chunk)
    constexpr int NSUB   = (GS > 0) ? 4 / GS : 1;  // quantisation groups per chunk (per-channel: 1 pseudo group)
    constexpr int SPG    = 4 / NSUB;               // MFMA k-steps per group
    constexpr bool QUANT = WBITS != 16;
    constexpr bool GROUPED = GS > 0;               // per-group scale (else one scale per column, applied at the end)
    constexpr bool CSIDE = QUANT && (MB <= 2 || BF); // zero/scale on the accumulator side (cheap for few row blocks);
                                                   // MB >= 3: classic operand-side dequant, cost independent of MB
    constexpr int GT     = 64 * NWN;               // threads per k-group
    constexpr int XSLOTS = 256 * MB;               // 16-byte slots per x chunk tile
    constexpr int UPT    = (XSLOTS + GT - 1) / GT; // x pieces per thread per chunk
    constexpr int XR     = D;                      // x ring depth: x(c) is issued before w(c), so the in-order
                                                   // vmcnt wait for x never drains newer weight loads
    constexpr int PPG    = 16 / NSUB;              // 16-byte pieces of a row per group
    __shared__ u32x4 xs[KG][2][XSLOTS];
    __shared__ float xsum[CSIDE ? KG : 1][2][NSUB * 2][16 * MB];
    __shared__ f32x4 red[(KG > 1) ? NWN * NBW * MB * 64 : 1];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MI355_TUNING
    unsigned long long stp[5] = {0, 0, 0, 0, 0};
    if (p.stamps) stp[0] = wall_clock64();
#define WQ_STAMP(i) do { if (p.stamps) stp[i] = wall_clock64(); } while (0)
#else
#define WQ_STAMP(i) do { } while (0)
#endif
    const int wn   = wave % NWN;                   // n-wave index
    const int kg   = wave / NWN;                   // k-group
    const int tg   = tid - kg * GT;                // thread index inside the k-group
    const int jj = lane & 15, q = lane >> 4;

    const int nt_base = (blockIdx.x * NWN + wn) * NBW;
    const int blk_c0  = blockIdx.y * p.cps;
    const int blk_nch = min(p.cps, p.KC - blk_c0);
    const int n_it    = (blk_nch + KG - 1) / KG;   // iterations = chunks of the longest k-group
    const int c_begin = blk_c0 + kg * n_it;
    const int n_ch    = max(0, min(n_it, blk_c0 + blk_nch - c_begin));

    // Bounds do the tail handling: every tile gets its own buffer descriptor covering exactly the chunks of
    // this k-group, so loads past its end (ring prefetch, short ranges) return 0 without memory traffic and
    // the main loop carries no conditional loads (a uniform select makes hipcc branch around the load and
    // drain vmcnt — measured: it serialised the whole ring).
    constexpr uint32_t FLAGS = 0x00020000u;
    __amdgpu_buffer_rsrc_t rw[NBW], rm[NBW];
    bool tile_ok[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int nt = nt_base + nb;
        tile_ok[nb]  = nt < p.NT;
        const char* wb = (const char*)p.qw + ((size_t)nt * p.KC + c_begin) * (LPC * 1024);
        rw[nb] = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, tile_ok[nb] ? n_ch * LPC * 1024 : 0, FLAGS);
        const char* mb = (const char*)p.meta + ((size_t)(GROUPED ? c_begin * NSUB : 0) * p.N_pad + nt * 16) * 4;
        const int mbytes = GROUPED ? (n_ch > 0 ? ((n_ch * NSUB - 1) * p.N_pad + 16) * 4 : 0) : 64;
        rm[nb] = __builtin_amdgcn_make_buffer_rsrc((void*)mb, 0, (tile_ok[nb] && QUANT) ? mbytes : 0, FLAGS);
    }
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, FLAGS);
    const uint32_t OOBX = 0x80000000u; // x images are < 2 GiB: stays out of range after adding chunk offsets

    u32x4    wr[D][NBW][LPC];
    uint32_t mr[D][NBW][NSUB];                     // meta {zneg, scale} of column 16nt + jj, per group
    f32x4    acc[NBW][MB];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint32_t lane16 = lane * 16u, jj4 = jj * 4u;
    const W4Consts w4c = w4_consts();
    auto load_w = [&](int d, int ci) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
#pragma unroll
            for (int lp = 0; lp < LPC; ++lp)
                wr[d][nb][lp] = bload128<2 /*nt*/>(rw[nb], lane16, (uint32_t)(ci * LPC + lp) * 1024u);
            if (GROUPED) {
#pragma unroll
                for (int gi = 0; gi < NSUB; ++gi)
                    mr[d][nb][gi] = __builtin_amdgcn_raw_buffer_load_b32(
                        rm[nb], jj4, (uint32_t)(ci * NSUB + gi) * (uint32_t)p.N_pad * 4u, 0);
            }
        }
    };

    // x staging: the GT threads of a k-group own the 256*MB 16-byte pieces of its [16*MB rows][128 k] tile
    uint32_t xoff[UPT];
    int      xlim[UPT]; // number of chunks of this k-group for which the piece is inside [0, K) and the group's range
    int      xslot[UPT], xsidx[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
        const int pi = tg + GT * u;
        const int j = pi >> 4, pp = pi & 15;
        const int k0 = c_begin * 128 + pp * 8;
        const bool ok = pi < XSLOTS && j < p.M && k0 < p.K;
        xoff[u]  = (uint32_t)((j * p.K + k0) * 2);
        xlim[u]  = ok ? min((p.K - k0 + 127) / 128, n_ch) : 0; // also zero past this k-group's range: OOB codes are not zero weights
        xslot[u] = (pi < XSLOTS) ? pp * (16 * MB) + (j ^ (pp & 3)) : -1;
        xsidx[u] = (pi < XSLOTS && (pp % PPG) == 0) ? (pp / PPG) * 2 * (16 * MB) + j : -1;
    }
    // x register ring: chunk c lives in xr[c % XR], loaded XR-1 iterations ahead, written to LDS one ahead
    u32x4 xr[XR][UPT];
    auto load_x = [&](int d, int ci) {
#pragma unroll
        for (int u = 0; u < UPT; ++u)   // per-lane select (v_cndmask): pieces past K / rows past M read as zero
            xr[d][u] = bload128<0>(rx, (ci < xlim[u]) ? xoff[u] : OOBX, (uint32_t)ci * 256u);
    };
    auto store_x = [&](int d, int buf) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            if (UPT * GT == XSLOTS || xslot[u] >= 0) xs[kg][buf][xslot[u]] = xr[d][u];
            if (CSIDE) {
                // group sums of x over the two code-bias classes; a row's 16 pieces sit in 16 adjacent lanes
                const f16x2 ones = {(f16)1.f, (f16)1.f};
                const u32x4 v = xr[d][u];
                float s0, s1;
                if (BF) {           // one code bias (128) for every position of the group
                    s0 = act_dot_ones<true>(v[0], 0.f); s0 = act_dot_ones<true>(v[1], s0);
                    s0 = act_dot_ones<true>(v[2], s0); s0 = act_dot_ones<true>(v[3], s0);
                    s1 = 0.f;
                } else if (WBITS == 4) {
                    s0 = __builtin_amdgcn_fdot2(as_h2(v[0]), ones, 0.f, false);
                    s1 = __builtin_amdgcn_fdot2(as_h2(v[1]), ones, 0.f, false);
                    s0 = __builtin_amdgcn_fdot2(as_h2(v[2]), ones, s0, false);
                    s1 = __builtin_amdgcn_fdot2(as_h2(v[3]), ones, s1, false);
                } else {
                    s0 = __builtin_amdgcn_fdot2(as_h2(v[0]), ones, 0.f, false);
                    s0 = __builtin_amdgcn_fdot2(as_h2(v[1]), ones, s0, false);
                    s0 = __builtin_amdgcn_fdot2(as_h2(v[2]), ones, s0, false);
                    s0 = __builtin_amdgcn_fdot2(as_h2(v[3]), ones, s0, false);
                    s1 = 0.f;
                }
                s0 = dpp_add<0xB1>(s0); s0 = dpp_add<0x4E>(s0);           // quad: xor 1, xor 2
                if (PPG >= 8) s0 = dpp_add<0x141>(s0);                     // row_half_mirror: 8 lanes
                if (PPG >= 16) s0 = dpp_add<0x140>(s0);                    // row_mirror: 16 lanes
                if (WBITS == 4 && !BF) {
                    s1 = dpp_add<0xB1>(s1); s1 = dpp_add<0x4E>(s1);
                    if (PPG >= 8) s1 = dpp_add<0x141>(s1);
                    if (PPG >= 16) s1 = dpp_add<0x140>(s1);
                }
                if (xsidx[u] >= 0) {
                    (&xsum[kg][buf][0][0])[xsidx[u]] = s0;
                    (&xsum[kg][buf][0][0])[xsidx[u] + 16 * MB] = s1;
                }
            }
        }
    };

    uint32_t mch[NBW]; // per-channel mode, operand-side path: the zero term is constant along K
    if (QUANT && !GROUPED && !CSIDE) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) mch[nb] = __builtin_amdgcn_raw_buffer_load_b32(rm[nb], jj4, 0, 0);
    }

    // prologue: x first, then weights (issue order matters for the in-order vmcnt)
#pragma unroll
    for (int d = 0; d < XR - 1; ++d) load_x(d, d);
#pragma unroll
    for (int d = 0; d < D; ++d) load_w(d, d);
    store_x(0, 0);
    __syncthreads();
    WQ_STAMP(1);

    auto compute = [&](int d, int buf) {
#pragma unroll
        for (int gi = 0; gi < NSUB; ++gi) {
            float XS[MB];
            f32x4 ag[NBW][MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                float xb = 0.f;
                if (CSIDE) {
                    const float x0 = xsum[kg][buf][gi * 2 + 0][mb * 16 + jj];
                    const float x1 = xsum[kg][buf][gi * 2 + 1][mb * 16 + jj];
                    XS[mb] = x0 + x1;
                    xb = BF ? 0.f : 960.f * x1;
                }
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) ag[nb][mb] = (f32x4){xb, xb, xb, xb};
            }
#pragma unroll
            for (int ss = 0; ss < SPG; ++ss) {
                const int s = gi * SPG + ss;
                u32x4 b[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) b[mb] = xs[kg][buf][(s * 4 + q) * (16 * MB) + mb * 16 + (jj ^ q)];
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    u32x4 a;
                    if (WBITS == 16) {
                        a = wr[d][nb][s % LPC];
                    } else if (CSIDE) {
                        if (WBITS == 4) {
                            if (BF) a = widen_w4_bf16(wr[d][nb][0][s]);
                            else    a = __builtin_bit_cast(u32x4, widen_w4(wr[d][nb][0][s], w4c));
                        } else {
                            const u32x4 w = wr[d][nb][(s >> 1) % LPC];
                            if (BF) a = widen_u8_bf16(w[(s & 1) * 2], w[(s & 1) * 2 + 1]);      // the stored byte itself: no code bias
                            else    a = __builtin_bit_cast(u32x4, widen_w8(w[(s & 1) * 2], w[(s & 1) * 2 + 1]));
                        }
                    } else { // operand-side dequant: (code - z) [* scale] in fp16, exact subtract, one rounding
                        const uint32_t m = GROUPED ? mr[d][nb][gi] : mch[nb];
                        const f16x2 zneg2 = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u));
                        const f16x2 sc2   = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
                        if (WBITS == 4) {
                            const f16x2 c960 = {(f16)960.f, (f16)960.f};
                            a = __builtin_bit_cast(u32x4, dequant_w4_vc(wr[d][nb][0][s], zneg2, zneg2 + c960, sc2, w4c));
                        } else {
                            const u32x4 w = wr[d][nb][(s >> 1) % LPC];
                            a = __builtin_bit_cast(u32x4, dequant_w8<GROUPED>(w[(s & 1) * 2], w[(s & 1) * 2 + 1], zneg2, sc2));
                        }
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        if (CSIDE) ag[nb][mb] = mfma_act<BF>(a, b[mb], ag[nb][mb]);
                        else       acc[nb][mb] = mfma_act<BF>(a, b[mb], acc[nb][mb]);
                    }
                }
            }
            // ---- C side: rows of this lane are columns 16nt + 4q + r; their meta sits in lanes 4q + r
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                if (!CSIDE) {
                    // operand-side path accumulated straight into acc
                } else if (GROUPED) {
                    f16x2 m4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        m4[r] = as_h2((uint32_t)__builtin_amdgcn_ds_bpermute((q * 4 + r) * 4, (int)mr[d][nb][gi]));
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // meta holds zneg = -(1024 + z); the bf16 codes are biased by 128
                            // (W8 bytes enter the bf16 MFMA unbiased: + 1024)
                            const float t = __builtin_fmaf(BF ? (float)m4[r][0] + (WBITS == 4 ? 896.f : 1024.f) : (float)m4[r][0], XS[mb], ag[nb][mb][r]);
                            acc[nb][mb][r] = __builtin_fmaf((float)m4[r][1], t, acc[nb][mb][r]);
                        }
                } else { // per-channel int8: zero code 128 for every column, scale applied in the epilogue
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[nb][mb][r] += __builtin_fmaf(BF ? -128.f : -1152.f, XS[mb], ag[nb][mb][r]);
                }
            }
        }
    };

    // Operand-side path (M > 32), software-pipelined by hand: the dequant of unit u+1 (13 VALU) is issued
    // between the 4 MFMAs of unit u, so MFMA and VALU overlap inside one wave (a wave issues in order: four
    // back-to-back MFMAs would otherwise block its own VALU for 64 cycles; PMC showed VALU 33% / MFMA 33% busy).
    auto compute_os = [&](int d, int buf) {
        constexpr int NS = 4 * NBW;
        f16x2 zn[NBW][NSUB], znb[NBW][NSUB], scl[NBW][NSUB];
        const f16x2 c960 = {(f16)960.f, (f16)960.f};
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int gi = 0; gi < NSUB; ++gi) {
                const uint32_t m = GROUPED ? mr[d][nb][gi] : mch[nb];
                zn[nb][gi]  = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u));
                scl[nb][gi] = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
                znb[nb][gi] = zn[nb][gi] + c960;
            }
        auto dq = [&](int u) -> f16x8 {
            const int s = u / NBW, nb = u % NBW, gi = s / SPG;
            if (WBITS == 4) return dequant_w4_vc(wr[d][nb][0][s], zn[nb][gi], znb[nb][gi], scl[nb][gi], w4c);
            const u32x4 w = wr[d][nb][(s >> 1) % LPC];
            return dequant_w8<GROUPED>(w[(s & 1) * 2], w[(s & 1) * 2 + 1], zn[nb][gi], scl[nb][gi]);
        };
        auto load_b = [&](f16x8 (&b)[MB], int s) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                b[mb] = __builtin_bit_cast(f16x8, xs[kg][buf][(s * 4 + q) * (16 * MB) + mb * 16 + (jj ^ q)]);
        };
        f16x8 b0[MB], b1[MB];
        load_b(b0, 0);
        f16x8 a_cur = dq(0), a_next = a_cur;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int s = u / NBW, nb = u % NBW;
            if (nb == 0 && s + 1 < 4) { if (s & 1) load_b(b0, s + 1); else load_b(b1, s + 1); }
            if (u + 1 < NS) a_next = dq(u + 1);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[nb][mb] = mfma16x16x32(a_cur, (s & 1) ? b1[mb] : b0[mb], acc[nb][mb]);
            a_cur = a_next;
        }
#pragma unroll
        for (int g = 0; g < NS * MB; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); // 4 VALU
        }
    };

    auto slot = [&](int d, int ci) {
        load_x((d + XR - 1) % XR, ci + XR - 1);  // OOB -> zeros past the end
        if (QUANT && !CSIDE) compute_os(d, d & 1);
        else compute(d, d & 1);                  // D is even, so ci & 1 == d & 1
        load_w(d, ci + D);
        store_x((d + 1) % XR, (d + 1) & 1);       // chunk ci+1, issued XR-2 iterations ago
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);         // keep slots apart: without it the scheduler hoists the widening
                                                   // of all D ring entries to the top of the round and spills
    };
    // Rounds of D slots with NO per-slot guard (n_it is rounded up; slots past the end see zero x and OOB = 0
    // weights).  With a guard inside the unrolled body hipcc's waitcnt pass must assume later slots may be
    // skipped and emits vmcnt(1) for every ring read: each iteration then waits for a load issued one
    // iteration earlier (measured: ~1 us per chunk).  Unguarded, the ring reads wait with vmcnt(3D - 4).
    for (int it = 0; it < n_it; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) slot(d, it + d);
    }

    WQ_STAMP(2);
    // ------------------------------------------------------------ merge k-groups, epilogue
    if (KG > 1) {
        if (kg > 0) {
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) red[((wn * NBW + nb) * MB + mb) * 64 + lane] = acc[nb][mb];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[nb][mb] += red[((wn * NBW + nb) * MB + mb) * 64 + lane];
    }
    __amdgpu_buffer_rsrc_t rp_slab = slab_rsrc(p, gridDim.y);
    __amdgpu_buffer_rsrc_t ry_f32 = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.mode == MODE_F32 ? (int)min((size_t)0x7FFFFFF0, (size_t)p.M * p.ldy * 4) : 0, FLAGS);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        if (!tile_ok[nb]) continue;
        const int n0 = (nt_base + nb) * 16 + q * 4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f};
        if (QUANT && !GROUPED) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = __builtin_amdgcn_raw_buffer_load_b32(rm[nb], (uint32_t)(q * 4 + r) * 4u, 0, 0);
                sc[r] = (float)as_h2(m)[1];
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = mb * 16 + jj;
            if (m >= p.M) continue;
            f32x4 v = acc[nb][mb] * sc;
            if (p.mode == MODE_PARTIAL) {
                // write-through (sc1): the slab leaves the L2 while the block's other waves / the other blocks still compute, instead
                // of as one dirty-line write-back of 6-14 MB at the end of the launch (measured M = 64, GEMM + consumer pair:
                // qkv 15.4 -> 14.5 us, o 14.4 -> 13.75, down 29.0 -> 27.1; nt and sc0 sc1 within 0.5 us of it)
                st_slab(rp_slab, (uint32_t)((((size_t)blockIdx.y * p.M + m) * p.N_pad + n0) * 4), v);
            } else {
                if (n0 >= p.N) continue;
                if (p.bias) {
                    const u32x2 bv = *reinterpret_cast<const u32x2*>(p.bias + n0);
                    v[0] += act_lo<BF>(bv[0]); v[1] += act_hi<BF>(bv[0]); v[2] += act_lo<BF>(bv[1]); v[3] += act_hi<BF>(bv[1]);
                }
                if (p.mode == MODE_F32) {   // lm_head logits (39 MB at b = 64): write-through like the slabs, nothing left dirty at the end
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry_f32, (uint32_t)(((size_t)m * p.ldy + n0) * 4), 0, 16 /*sc1*/);
                } else if (p.mode == MODE_F16) {
                    u32x2 o;
                    if (p.y_img) {   // activation image (common.h): fp16 of the BF-typed tensor's values (x 2^-8 for bf16)
                        o[0] = act_pack<false>(img_val<BF>(act_round<BF>(v[0])), img_val<BF>(act_round<BF>(v[1])));
                        o[1] = act_pack<false>(img_val<BF>(act_round<BF>(v[2])), img_val<BF>(act_round<BF>(v[3])));
                        *reinterpret_cast<u32x2*>((f16*)p.y + act_img_index(m, n0, (p.M + 15) >> 4)) = o;
                    } else {
                        o[0] = act_pack<BF>(v[0], v[1]); o[1] = act_pack<BF>(v[2], v[3]);
                        *reinterpret_cast<u32x2*>((f16*)p.y + (size_t)m * p.ldy + n0) = o;
                    }
                } else { // MODE_SILU: (gate, up) interleaved; round GEMM output to the activation dtype first
                    float sg[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float g = act_round<BF>(v[2 * t]), u = act_round<BF>(v[2 * t + 1]);
                        sg[t] = (g / (1.f + __expf(-g))) * u;
                    }
                    if (p.y_img)
                        *reinterpret_cast<uint32_t*>((f16*)p.y + act_img_index(m, n0 >> 1, (p.M + 15) >> 4)) = act_pack<false>(img_val<BF>(act_round<BF>(sg[0])), img_val<BF>(act_round<BF>(sg[1])));
                    else
                        *reinterpret_cast<uint32_t*>((f16*)p.y + (size_t)m * p.ldy + (n0 >> 1)) = act_pack<BF>(sg[0], sg[1]);
                }
            }
        }
    }
#ifdef MI355_TUNING
    if (p.stamps) {
        stp[3] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stp[4] = wall_clock64();
        if (tid == 0) {
            unsigned long long* d = p.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 5;
            for (int i = 0; i < 5; ++i) d[i] = stp[i];
        }
    }
#endif
#undef WQ_STAMP
}

// Sum split-K slabs (+bias) and apply the epilogue.  One thread per 4 columns.
template <bool BF>
__global__ __launch_bounds__(256) void reduce_epilogue_kernel(const float* __restrict__ partials, int nsplit,
                                                              int M, int N, int N_pad, const f16* __restrict__ bias,
                                                              void* y, int ldy, int mode, int y_img_mblk) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n4  = N_pad >> 2;
    if (idx >= M * n4) return;
    const int m = idx / n4, n0 = (idx - m * n4) * 4;
    if (n0 >= N) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s)
        v += *reinterpret_cast<const f32x4*>(partials + ((size_t)s * M + m) * N_pad + n0);
    if (bias) {
        const u32x2 bv = *reinterpret_cast<const u32x2*>(bias + n0);
        v[0] += act_lo<BF>(bv[0]); v[1] += act_hi<BF>(bv[0]); v[2] += act_lo<BF>(bv[1]); v[3] += act_hi<BF>(bv[1]);
    }
    if (mode == MODE_F32) {
        *reinterpret_cast<f32x4*>((float*)y + (size_t)m * ldy + n0) = v;
    } else if (mode == MODE_F16) {
        u32x2 o;
        if (y_img_mblk > 0) {   // activation image (common.h): fp16, the values of the BF-typed tensor (x 2^-8 for bf16) at the image's addresses
            o[0] = act_pack<false>(img_val<BF>(act_round<BF>(v[0])), img_val<BF>(act_round<BF>(v[1])));
            o[1] = act_pack<false>(img_val<BF>(act_round<BF>(v[2])), img_val<BF>(act_round<BF>(v[3])));
            *reinterpret_cast<u32x2*>((f16*)y + act_img_index(m, n0, y_img_mblk)) = o;
        } else {
            o[0] = act_pack<BF>(v[0], v[1]); o[1] = act_pack<BF>(v[2], v[3]);
            *reinterpret_cast<u32x2*>((f16*)y + (size_t)m * ldy + n0) = o;
        }
    } else {
        float sg[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float g = act_round<BF>(v[2 * t]), u = act_round<BF>(v[2 * t + 1]);
            sg[t] = (g / (1.f + __expf(-g))) * u;
        }
        if (y_img_mblk > 0)
            *reinterpret_cast<uint32_t*>((f16*)y + act_img_index(m, n0 >> 1, y_img_mblk)) = act_pack<false>(img_val<BF>(act_round<BF>(sg[0])), img_val<BF>(act_round<BF>(sg[1])));
        else
            *reinterpret_cast<uint32_t*>((f16*)y + (size_t)m * ldy + (n0 >> 1)) = act_pack<BF>(sg[0], sg[1]);
    }
}

void launch_reduce_epilogue(const float* partials, int nsplit, int M, int N, int N_pad, const f16* bias, void* y, int ldy, int mode, bool bf,
                            hipStream_t st, int y_img_mblk = 0) {
    const int total = M * (N_pad / 4);
    if (bf) hipLaunchKernelGGL(reduce_epilogue_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, st, partials, nsplit, M, N, N_pad, bias, y, ldy, mode, y_img_mblk);
    else    hipLaunchKernelGGL(reduce_epilogue_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, st, partials, nsplit, M, N, N_pad, bias, y, ldy, mode, y_img_mblk);
}

// ------------------------------------------------------------------ dispatch
struct GemmPlan { int cfg, nsplit, cps, bn; };   // cfg: index into the block-shape table below

// Tuning switches exist only in the MI355_TUNING build (lib/libmi355_decode_tuning.so, used by tools/): the product library
// has no mutable process-global state.  [1] nsplit override, [2] config override (+1), [4] == 2: gemm_smallm also inside
// the decode step.  Per-call kernel-family hints travel in the `epilogue` word instead (MI355_HINT_*).

// block shapes: {MB, NBW, NWN, KG}
//   cfg 0: M<=16, BN=64   (4 n-waves x 2 k-groups)      cfg 1: M<=16, BN=128 (huge N, e.g. lm_head)
//   cfg 2: M<=32, BN=128  (8 n-waves)                   cfg 3: M<=48, BN=256   cfg 4: M<=64, BN=256
//   cfg 5: M<=16, BN=64, 4 waves, no k-groups (experiment)   cfg 6/7: M<=64/48, BN=256 as 16 n-waves x 1 tile
//   cfg 8: M<=64 BN=128 8x1 tiles   cfg 9: M<=64 BN=128 4 waves x 2 tiles   cfg 10: M<=64 BN=256 4 waves x 4 tiles
//   cfg 11: M<=64 BN=160 as 10 n-waves (237 gate_up blocks on 256 CUs)
//   cfg 12: M<=64 BN=64 as 4 waves x 1 tile (592 gate_up blocks, several resident per CU)
constexpr int kCfgBN[13] = {64, 128, 128, 256, 256, 64, 256, 256, 128, 128, 256, 160, 64};

// Ring depths per shape: D1 for the M<=16 shapes, D2 for BN=128 (M<=32), DW for the 16-wave BN=256 shapes.
// Bytes in flight per CU = waves * NBW * D KiB; ~85 KiB per CU are needed to cover HBM latency at 6 TB/s.
template <int WBITS, int GS, int D1, int D2, int DW>
int launch_gemm_t(const GemmParams& p, int cfg, hipStream_t st) {
    dim3 grid(cdiv(p.NT * 16, kCfgBN[cfg]), p.nsplit);
    if (cfg == 3) cfg = 4;   // (the three-row-block shapes 3 / 7 are gone: 33..48 rows run on the four-row-block ones)
    if (cfg == 7) cfg = 6;
    switch (cfg) {
        case 0: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 1, 1, GS, D1, 4, 2>), grid, dim3(512), 0, st, p); break;
        case 1: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 1, 2, GS, D1, 4, 2>), grid, dim3(512), 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 2, 1, GS, D2, 8, 1>), grid, dim3(512), 0, st, p); break;
        case 4: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 2, GS, DW, 8, 1>), grid, dim3(512), 0, st, p); break;
        case 5: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 1, 1, GS, D1, 4, 1>), grid, dim3(256), 0, st, p); break;
        case 6: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 1, GS, 2, 16, 1>), grid, dim3(1024), 0, st, p); break;
        case 8: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 1, GS, DW, 8, 1>), grid, dim3(512), 0, st, p); break;
        case 9: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 2, GS, DW, 4, 1>), grid, dim3(256), 0, st, p); break;
        case 10: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 4, GS, 2, 4, 1>), grid, dim3(256), 0, st, p); break;
        case 11: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 1, GS, DW, 10, 1>), grid, dim3(640), 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 1, GS, DW, 4, 1>), grid, dim3(256), 0, st, p); break;
    }
    MI355_CHECK_LAUNCH("gemm_wq_kernel");
    return MI355_OK;
}

// bf16 activations: the three shapes the planner picks by row-block count (+ the narrow-N shape 8), nothing else instantiated
template <int WBITS, int GS, int D1, int D2, int DW>
int launch_gemm_bf16_t(const GemmParams& p, int cfg, hipStream_t st) {
    dim3 grid(cdiv(p.NT * 16, kCfgBN[cfg]), p.nsplit);
    switch (cfg) {
        case 1: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 1, 2, GS, D1, 4, 2, true>), grid, dim3(512), 0, st, p); break;
        case 5: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 1, 1, GS, D1, 4, 1, true>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 2, 1, GS, D2, 8, 1, true>), grid, dim3(512), 0, st, p); break;
        case 8: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 1, GS, DW, 8, 1, true>), grid, dim3(512), 0, st, p); break;
        case 6: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 1, GS, 2, 16, 1, true>), grid, dim3(1024), 0, st, p); break;
#ifdef MI355_TUNING   // other shapes tried for the four-row-block bf16 case (tools/gemm_bench.py --bf16 1 --nbw N): all behind 6 / 8
        case 9: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 2, GS, DW, 4, 1, true>), grid, dim3(256), 0, st, p); break;
        case 10: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 4, GS, 2, 4, 1, true>), grid, dim3(256), 0, st, p); break;
        case 12: hipLaunchKernelGGL((gemm_wq_kernel<WBITS, 4, 1, GS, DW, 4, 1, true>), grid, dim3(256), 0, st, p); break;
#endif
        default: mi355_set_error("gemm (bf16): block shape %d not built", cfg); return MI355_ERR_UNSUPPORTED;
    }
    MI355_CHECK_LAUNCH("gemm_wq_kernel<bf16>");
    return MI355_OK;
}

int launch_gemm(const GemmParams& p, int wbits, int group_size, int cfg, hipStream_t st) {
    if (p.bf16) {
        if (wbits == 16) return launch_gemm_bf16_t<16, 0, 2, 2, 2>(p, cfg, st);
        if (wbits == 4 && group_size == 128) return launch_gemm_bf16_t<4, 4, 4, 4, 4>(p, cfg, st);
        if (wbits == 4 && group_size == 64) return launch_gemm_bf16_t<4, 2, 4, 4, 4>(p, cfg, st);
        if (wbits == 4 && group_size == 32) return launch_gemm_bf16_t<4, 1, 4, 4, 4>(p, cfg, st);
        if (wbits == 8 && group_size == 0) return launch_gemm_bf16_t<8, 0, 4, 2, 2>(p, cfg, st);
        if (wbits == 8 && group_size == 128) return launch_gemm_bf16_t<8, 4, 4, 2, 2>(p, cfg, st);
        mi355_set_error("gemm: bf16 activations: unsupported wbits=%d group_size=%d", wbits, group_size);
        return MI355_ERR_UNSUPPORTED;
    }
    if (wbits == 16) return launch_gemm_t<16, 0, 2, 2, 2>(p, cfg, st);
    if (wbits == 4) {
#ifdef MI355_TUNING
        if (group_size == 128 && TUNE(3) == 1) return launch_gemm_t<4, 4, 8, 4, 4>(p, cfg, st);     // ring depth experiment
        if (group_size == 128 && TUNE(3) == 2) return launch_gemm_t<4, 4, 12, 4, 4>(p, cfg, st);
#endif
        if (group_size == 128) return launch_gemm_t<4, 4, 4, 4, 4>(p, cfg, st);
        if (group_size == 64) return launch_gemm_t<4, 2, 4, 4, 4>(p, cfg, st);
        if (group_size == 32) return launch_gemm_t<4, 1, 4, 4, 4>(p, cfg, st);
    }
    if (wbits == 8) {
        if (group_size == 0) return launch_gemm_t<8, 0, 4, 2, 2>(p, cfg, st);
        if (group_size == 128) return launch_gemm_t<8, 4, 4, 2, 2>(p, cfg, st);
    }
    mi355_set_error("gemm: unsupported wbits=%d group_size=%d", wbits, group_size);
    return MI355_ERR_UNSUPPORTED;
}

int check_weight(const mi355_weight_t* w) {
    MI355_CHECK_ARG(w && w->qweight, "linear: null weight");
    MI355_CHECK_ARG(w->wbits == 4 || w->wbits == 8 || w->wbits == 16, "linear: wbits=%d", w->wbits);
    MI355_CHECK_ARG(w->K > 0 && w->N > 0 && w->K % 8 == 0 && w->N % 8 == 0, "linear: K=%d N=%d must be multiples of 8", w->K, w->N);
    MI355_CHECK_ARG(w->K_pad % 128 == 0 && w->K_pad >= w->K, "linear: K_pad=%d", w->K_pad);
    MI355_CHECK_ARG(w->N_pad % 16 == 0 && w->N_pad >= w->N, "linear: N_pad=%d", w->N_pad);
    MI355_CHECK_ARG(w->wbits == 16 || w->meta, "linear: quantized weight needs meta");
    MI355_CHECK_ARG((uint64_t)w->K_pad * w->N_pad * w->wbits / 8 < 0xFFFFFFF0ull, "linear: weight image >= 4 GiB");
    MI355_CHECK_ARG(w->act_dtype == MI355_ACT_F16 || w->act_dtype == MI355_ACT_BF16, "linear: act_dtype=%d", w->act_dtype);
    return MI355_OK;
}

void fill_params(GemmParams& p, const void* x, int M, const mi355_weight_t* w) {
    p.x = (const f16*)x; p.qw = w->qweight; p.meta = (const uint32_t*)w->meta;
    p.M = M; p.K = w->K; p.N = w->N; p.N_pad = w->N_pad; p.NT = w->N_pad / 16; p.KC = w->K_pad / 128;
    p.qw_bytes = (uint32_t)((uint64_t)w->K_pad * w->N_pad * w->wbits / 8);
    const int ngroups = (w->wbits == 16) ? 0 : (w->group_size > 0 ? w->K_pad / w->group_size : 1);
    p.meta_bytes = (uint32_t)((uint64_t)ngroups * w->N_pad * 4);
    p.x_bytes = (uint32_t)((uint64_t)M * w->K * 2);
    p.bias = nullptr; p.y = nullptr; p.partials = nullptr; p.ldy = 0;
    p.bf16 = w->act_dtype == MI355_ACT_BF16; p.x_img = 0; p.y_img = 0;
#ifdef MI355_TUNING
    p.stamps = (TUNE(7) == 2) ? g_wide_stamps : nullptr;
#endif
}

// Block shape + split-K plan.  The kernel is a per-block pipeline of n_it chunk iterations (one barrier each);
// with few blocks the iteration latency, not bandwidth, sets the time, so K is split until the machine is
// full.  Split-K costs fp32 slab traffic (8 B per output element per split, write + read).  Pick the split
// that minimises a small time model calibrated on MI355X (profiles/r01_*):
//     t = ceil(blocks / resident) * n_it * t_it  +  slab_bytes / 3 TB/s
GemmPlan plan_gemm(int M, const mi355_weight_t* w, int max_splits) {
    const int NT = w->N_pad / 16, KC = w->K_pad / 128;
    // 33..48 rows run on the four-row-block shapes: the three-row-block instances measured behind them (round 3 batch sweep:
    // a step at b = 33..48 took 3.66-3.71 ms against 3.48 ms at b = 64; rows past M are zero activations)
    const int MB = cdiv(M, 16) == 3 ? 4 : cdiv(M, 16);
    GemmPlan g;
    g.cfg = (MB == 1) ? (NT >= 4096 ? 1 : 5) : (MB == 2 ? 2 : 4); // measured best per row-block count
    // bf16, four row blocks (accumulator-side dequant): one tile per wave fits the registers; 16 waves x BN = 256 for the big
    // weights, 8 waves x BN = 128 for the short-K ones (measured M = 64, g128: gate_up 67.2 / 51.4 us, down 23.9 / 21.8 us as
    // shape 8 / 6; qkv 9.7 / 11.7, o 9.9 / 11.2: profiles/r03_bf16_gemm_m64_shapes.txt)
    if (w->act_dtype == MI355_ACT_BF16 && MB >= 3) g.cfg = ((uint64_t)w->K_pad * w->N_pad * w->wbits / 8 >= (16u << 20)) ? 6 : 8;
    if (TUNE(2) > 0) g.cfg = TUNE(2) - 1;
    g.bn = kCfgBN[g.cfg];
    const int blocks_n = cdiv(NT * 16, g.bn);
    static const double t_it_us[5]  = {0.0, 0.8, 1.0, 1.3, 1.5};   // per chunk iteration, by MB
    static const int    resident[13] = {2, 2, 2, 1, 1, 4, 1, 1, 1, 2, 2, 1, 3};     // blocks per CU, by shape
    const bool kgrouped = g.cfg <= 1;
    const int min_chunks = kgrouped ? 4 : 2;
    int best = 1; double best_t = 1e30;
    for (int ns = 1; ns <= max_splits && ns <= (KC / min_chunks > 0 ? KC / min_chunks : 1); ++ns) {
        const int cps = cdiv(KC, ns), nsr = cdiv(KC, cps);
        if (nsr != ns) continue;
        const int n_it = kgrouped ? cdiv(cps, 2) : cps;
        const int rounds = cdiv(blocks_n * ns, 256 * resident[g.cfg]);
        const double slab = ns > 1 ? (double)ns * M * w->N_pad * 8.0 / 3.0e6 : 0.0; // us at 3 TB/s
        const double t = rounds * (n_it * t_it_us[MB] + 2.0) + slab;
        if (t < best_t - 1e-9) { best_t = t; best = ns; }
    }
    int nsplit = best;
    if (TUNE(1) > 0) nsplit = TUNE(1) > max_splits ? max_splits : TUNE(1);
    g.cps    = cdiv(KC, nsplit);
    g.nsplit = cdiv(KC, g.cps);
    // Short-K / narrow-N shapes at M > 32 (qkv, o): BN = 256 leaves < 2/3 of the CUs with a block even after the
    // split; the BN = 128 shape (8 n-waves x 1 tile) doubles the block count at the same slab traffic
    // (measured M = 64: qkv 10.97 -> 9.45 us, o 10.91 -> 9.17 us; down stays on BN = 256, 21.0 vs 23.2 us).
    if (MB >= 3 && TUNE(2) == 0 && w->wbits != 16 && w->act_dtype == MI355_ACT_F16 && blocks_n * g.nsplit <= 160) {
        g.cfg = 8;
        g.bn  = kCfgBN[8];
    }
    // 16-bit weights with a vocabulary-sized N (lm_head) at M > 32: 1188 blocks of BN = 128 fill the rounds over the CUs better
    // than 594 of BN = 256 (measured M = 64, 152064 columns: 204.8 -> 187.9 us = 5.80 TB/s; profiles/r03_lmhead_block_shapes.txt)
    if (MB >= 3 && TUNE(2) == 0 && w->wbits == 16 && NT >= 4096 && w->act_dtype == MI355_ACT_F16) {
        g.cfg = 8;
        g.bn  = kCfgBN[8];
    }
    return g;
}

} // namespace

extern "C" int mi355_gemm_plan(int M, const mi355_weight_t* w, int max_splits, int* nbw_out, int* cps_out) {
    const GemmPlan g = plan_gemm(M, w, max_splits);
    if (nbw_out) *nbw_out = g.cfg;
    if (cps_out) *cps_out = g.cps;
    return g.nsplit;
}

#ifdef MI355_TUNING
// Tuning / experiment hook (tools/gemm_bench.py); only in the tuning build, not part of the ABI.
extern int g_wide_dbg;
extern "C" void mi355_debug_set(int key, int value) {
    if (key == 0) g_wide_dbg = value;
    if (key >= 0 && key < 16) g_tune[key] = value;
}
#endif

extern "C" size_t mi355_linear_workspace_bytes(int32_t M, const mi355_weight_t* w) {
    if (!w || M <= 0) return 0;
    const int Mc = M > 64 ? 64 : M;
    size_t need = 0;
    for (int m = 1; m <= Mc; ++m) { // forward() plans per slab of <= 64 rows; cover every possible slab height
        const GemmPlan g = plan_gemm(m, w, 64);
        const size_t b = g.nsplit > 1 ? (size_t)g.nsplit * m * w->N_pad * sizeof(float) : 0;
        need = b > need ? b : need;
    }
    return need;
}

extern "C" int mi355_linear_partial(const void* x, int32_t M, const mi355_weight_t* w, float* partials,
                                    int32_t max_splits, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x && partials && M > 0 && M <= 64 && max_splits >= 1, "linear_partial: bad args (M=%d)", M);
    GemmParams p; fill_params(p, x, M, w);
    p.mode = MODE_PARTIAL; p.partials = partials;
    // (gemm_smallm.hip is not used here: inside the decode step, where the consumer kernel folds the slabs anyway,
    // it measured 3-8 % behind the staged kernel -- b=1 linears 1.55 vs 1.48 ms/step; tuning switch 4 = 2 re-enables it)
    const bool bf = p.bf16;    // bf16 activations: the staged kernel is the only family built for them
    if (!bf && M <= 8 && w->wbits != 16 && TUNE(4) == 2) {
        const int rc = mi355_gemm_smallm(&p, w->wbits, w->group_size, 1, max_splits, stream);
        if (rc != MI355_ERR_UNSUPPORTED) return rc;
    }
    if (!bf && M > 16 && w->wbits != 16 && TUNE(5) != 1) { // register-resident activations, K split over the waves (gemm_wide.hip)
        const int rc = mi355_gemm_wide(&p, w->wbits, w->group_size, 1, max_splits, stream);
        if (rc != MI355_ERR_UNSUPPORTED) return rc;
    }
    const GemmPlan g = plan_gemm(M, w, max_splits);
    p.nsplit = g.nsplit; p.cps = g.cps;
    if (int e = launch_gemm(p, w->wbits, w->group_size, g.cfg, (hipStream_t)stream)) return e;
    return p.nsplit;
}

extern "C" int mi355_linear_forward(const void* x, int32_t M, const mi355_weight_t* w, const void* bias,
                                    void* y, int32_t epilogue, void* workspace, size_t workspace_bytes,
                                    mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x && y && M > 0, "linear_forward: bad args");
    const int mode = (epilogue & MI355_EPI_OUT_F32) ? MODE_F32 : (epilogue & MI355_EPI_SILU_MUL) ? MODE_SILU : MODE_F16;
    const int ldy  = (mode == MODE_SILU) ? w->N / 2 : w->N;
    const size_t ysz = (mode == MODE_F32) ? 4 : 2;
    hipStream_t st = (hipStream_t)stream;
    // prefill-sized M: the compute-shaped kernel reads every weight once per 128 rows.  Its grid is (N / 256) x (M / 128)
    // blocks without a K split: narrow outputs at moderate M (down_proj, N = 3584: 14 blocks per 128 rows) would leave
    // most CUs idle (measured M = 128: 246 us vs 2 x 21 us as 64-row slabs), so those stay on the decode kernels
    const bool bf = w->act_dtype == MI355_ACT_BF16;   // bf16 activations: 64-row slabs through the staged kernel at every M
    // compute-shaped kernel or 64-row slabs of the decode kernels?  A small time model (us), calibrated on MI355X at the Qwen2-7B
    // shapes (profiles/r03_prefill_gemm_tiles_per_wave.txt): the prefill kernel takes ~2.2 us per 128-k chunk per round of 256
    // column x row blocks, a 64-row slab ~(weight MB / 1.6 + 6) us.  Narrow outputs at moderate M stay on the slabs (down_proj
    // at M = 512: 56 blocks, 280 vs 218 us), wider / deeper cases move over earlier than the old ">= 128 blocks" rule allowed
    // (o_proj at M = 1024: 112 blocks, ~68 vs 204 us).
    const long pf_blocks = (long)cdiv(w->N_pad / 16, 16) * cdiv(M, 128);
    const double t_pf = (double)cdiv((int)pf_blocks, 256) * (w->K_pad / 128) * 2.2;
    const double t_slab = (double)cdiv(M, 64) * ((double)w->K_pad * w->N_pad * w->wbits / 8 / 1.6e6 + 6.0);
    if (!bf && M >= 128 && w->wbits != 16 && t_pf < t_slab) {
        GemmParams ps; fill_params(ps, x, M, w);
        ps.mode = mode; ps.bias = (const f16*)bias; ps.y = y; ps.ldy = ldy;
        const int rc = mi355_gemm_prefill(&ps, w->wbits, w->group_size, stream);
        if (rc != MI355_ERR_UNSUPPORTED) return rc;
    }
    for (int m0 = 0; m0 < M; m0 += 64) {
        const int Mc = (M - m0) > 64 ? 64 : (M - m0);
        GemmParams p; fill_params(p, (const f16*)x + (size_t)m0 * w->K, Mc, w);
        if (!bf && Mc <= 8 && w->wbits != 16 && !(epilogue & MI355_HINT_NO_PERSISTENT) && TUNE(4) != 1) { // persistent x-resident kernel: fused epilogue, no slabs, no
                                                             // reduce launch (stand-alone call: qkv 9.1 vs 14.6 us at M = 1)
            GemmParams ps; fill_params(ps, (const f16*)x + (size_t)m0 * w->K, Mc, w);
            ps.mode = mode; ps.bias = (const f16*)bias; ps.y = (char*)y + (size_t)m0 * ldy * ysz; ps.ldy = ldy;
            const int rc = mi355_gemm_smallm(&ps, w->wbits, w->group_size, 0, 1, stream);
            if (rc >= 0) continue;
            if (rc != MI355_ERR_UNSUPPORTED) return rc;
        }
        if (!bf && Mc > 16 && w->wbits != 16 && !(epilogue & MI355_HINT_STAGED) && TUNE(5) != 1) {
            GemmParams ps; fill_params(ps, (const f16*)x + (size_t)m0 * w->K, Mc, w);
            ps.mode = mode; ps.bias = (const f16*)bias; ps.y = (char*)y + (size_t)m0 * ldy * ysz; ps.ldy = ldy;
            int rc = MI355_ERR_UNSUPPORTED;
            if (rc == MI355_ERR_UNSUPPORTED) rc = mi355_gemm_wide(&ps, w->wbits, w->group_size, 0, 1, stream);
            if (rc >= 0) continue;
            if (rc != MI355_ERR_UNSUPPORTED) return rc;
        }
        const GemmPlan g = plan_gemm(Mc, w, 64);
        const int ns = g.nsplit, cps = g.cps;
        if (ns > 1 && (size_t)ns * Mc * w->N_pad * sizeof(float) > workspace_bytes) {
            mi355_set_error("linear_forward: workspace %zu too small", workspace_bytes);
            return MI355_ERR_WORKSPACE;
        }
        void* yc = (char*)y + (size_t)m0 * ldy * ysz;
        p.nsplit = ns; p.cps = cps; p.ldy = ldy;
        if (ns == 1) {
            p.mode = mode; p.bias = (const f16*)bias; p.y = yc;
            if (int e = launch_gemm(p, w->wbits, w->group_size, g.cfg, st)) return e;
        } else {
            p.mode = MODE_PARTIAL; p.partials = (float*)workspace;
            if (int e = launch_gemm(p, w->wbits, w->group_size, g.cfg, st)) return e;
            launch_reduce_epilogue((const float*)workspace, ns, Mc, w->N, w->N_pad, (const f16*)bias, yc, ldy, mode, bf, st);
            MI355_CHECK_LAUNCH("reduce_epilogue_kernel");
        }
    }
    return MI355_OK;
}

// Engine-internal: GEMM with fused epilogue.  Without a workspace: one launch, nsplit = 1.  With one (the step driver's
// slab buffer): shapes that would leave most CUs without a block in a single launch (column-parallel shards under TP:
// gate_up of Qwen2-7B at tp = 4 is 37-74 blocks) are split along K into slabs and finished by reduce_epilogue_kernel.
extern "C" int mi355_linear_direct(const void* x, int32_t M, const mi355_weight_t* w, const void* bias, void* y,
                                   int32_t epilogue, void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x && y && M > 0 && M <= 64, "linear_direct: bad args");
    const int mode = (epilogue & MI355_EPI_OUT_F32) ? MODE_F32 : (epilogue & MI355_EPI_SILU_MUL) ? MODE_SILU : MODE_F16;
    GemmParams p; fill_params(p, x, M, w);
    p.mode = mode; p.bias = (const f16*)bias; p.y = y;
    p.ldy = (mode == MODE_SILU) ? w->N / 2 : w->N;
    const bool bf = p.bf16;
    // MI355_EPI_OUT_IMAGE: the 16-bit output as an activation image (what a launch on images reads next: under TP the SiLU output of the
    // gate_up shard for down_proj's K-quarter launch) -- the kernels' common store and the slab fold below both know the layout
    const bool out_img = (epilogue & MI355_EPI_OUT_IMAGE) != 0;
    if (out_img) {
        MI355_CHECK_ARG(mode != MODE_F32 && p.ldy % 32 == 0 && !(bf && bias), "linear_direct: an image output is a 16-bit tensor with a multiple of 32 columns (bf16: no bias)");
        p.y_img = 1;
    }
    if (!bf && !out_img && M <= 8 && w->wbits != 16 && (TUNE(4) == 2 || TUNE(7) == 1)) {
        const int rc = mi355_gemm_smallm(&p, w->wbits, w->group_size, 0, 1, stream);
        if (rc >= 0) return MI355_OK;
        if (rc != MI355_ERR_UNSUPPORTED) return rc;
    }
    if (!bf && M > 16 && w->wbits != 16 && !(epilogue & MI355_HINT_STAGED) && TUNE(5) != 1) {
        int rc = MI355_ERR_UNSUPPORTED;
        if (rc == MI355_ERR_UNSUPPORTED) rc = mi355_gemm_wide(&p, w->wbits, w->group_size, 0, 1, stream);
        if (rc >= 0) return MI355_OK;
        if (rc != MI355_ERR_UNSUPPORTED) return rc;
    }
    int max_splits = 1;
    if (workspace && w->wbits != 16) {
        const size_t per_split = (size_t)M * w->N_pad * sizeof(float);
        const size_t fit = workspace_bytes / per_split;
        max_splits = fit > 16 ? 16 : (int)fit;
        if (max_splits < 1) max_splits = 1;
    }
    const GemmPlan g = plan_gemm(M, w, max_splits);
    p.nsplit = g.nsplit; p.cps = g.cps;
    if (g.nsplit == 1) return launch_gemm(p, w->wbits, w->group_size, g.cfg, (hipStream_t)stream);
    p.mode = MODE_PARTIAL; p.partials = (float*)workspace; p.bias = nullptr; p.y = nullptr;
    if (int e = launch_gemm(p, w->wbits, w->group_size, g.cfg, (hipStream_t)stream)) return e;
    launch_reduce_epilogue((const float*)workspace, g.nsplit, M, w->N, w->N_pad, (const f16*)bias, y, (mode == MODE_SILU) ? w->N / 2 : w->N, mode, bf,
                           (hipStream_t)stream, out_img ? cdiv(M, 16) : 0);
    MI355_CHECK_LAUNCH("reduce_epilogue_kernel");
    return MI355_OK;
}

// ------------------------------------------------------------------ full-K kernels with fused consumers (gemm_fullk.hip)
extern "C" int mi355_fullk_weight_ok(const mi355_weight_t* w) {
    if (w && w->act_dtype != MI355_ACT_F16) return 0;   // the full-K fused launches exist for fp16 activations only
    const bool fmt = w && w->qweight && ((w->wbits == 4 && w->meta && (w->group_size == 128 || w->group_size == 64 || w->group_size == 32)) || w->wbits == 16);
    return fmt && w->K % 128 == 0 && w->K_pad == w->K && w->K_pad / 128 >= 4 && w->N % 16 == 0;
}

// the launches on activation images (gemm_fullk64 / gemm_splitk64 / gemm_wide image entries): W4 group-wise weights in the native
// image; the activation dtype of the surrounding tensors is free (the image itself is fp16, see common.h)
static bool img_weight_ok(const mi355_weight_t* w) {
    const bool fmt = w && w->qweight && w->meta && ((w->wbits == 4 && (w->group_size == 128 || w->group_size == 64 || w->group_size == 32)) ||
                                                    (w->wbits == 8 && w->group_size == 0));   // round 5: per-channel W8 (load-time INT8 autoquant)
    return fmt && w->K % 128 == 0 && w->K_pad == w->K && w->K_pad / 128 >= 4 && w->N % 16 == 0;
}

// the predicate of the full-K image launches (gemm_fullk64.hip: <= 15 K-slice waves of <= 3 chunks), shared with decoder_create so
// that the step driver and the launchers cannot drift apart
extern "C" int mi355_fullk64_weight_ok(const mi355_weight_t* w) { return img_weight_ok(w) && w->K_pad / 128 <= (w->wbits == 8 ? 30 : 45); }
// ... of the QKV launch with its RoPE epilogue: beyond 45 chunks (hidden 8192) only the g128 instances of <= 2 row blocks per block
// exist, i.e. the row-split form: the tile pairs have to leave half of the 256 CUs free (a TP shard's do)
extern "C" int mi355_fullk64_qkv_ok(const mi355_weight_t* w, int32_t hd) {
    if (!img_weight_ok(w) || (hd != 64 && hd != 128) || w->N % hd != 0) return 0;
    const int KC = w->K_pad / 128;
    if (KC <= (w->wbits == 8 ? 30 : 45)) return 1;
    return w->wbits == 4 && w->group_size == 128 && KC <= 75 && 2 * (w->N / 32) <= 256;
}

// what block u of gemm_fullk64's QKV launch reads (internal.h: mi355_touch_t), for the spare blocks of the launch in front of it
extern "C" int mi355_qkv_touch_plan(const mi355_weight_t* wqkv, int32_t hd, void* sink, mi355_touch_t* out) {
    if (!out || !sink || !img_weight_ok(wqkv) || (hd != 64 && hd != 128) || wqkv->N % hd != 0) return MI355_ERR_UNSUPPORTED;
    const int KC = wqkv->K_pad / 128;
    out->qw = wqkv->qweight; out->meta = wqkv->meta;
    out->run_bytes = (uint32_t)KC * 256u * (uint32_t)wqkv->wbits;        // 1 KB (W4) / 2 KB (W8) per (tile, chunk), a tile's chunks back to back
    out->meta_groups = wqkv->group_size > 0 ? (uint32_t)(wqkv->K_pad / wqkv->group_size) : 1u; out->meta_stride = (uint32_t)wqkv->N_pad;
    out->n_units = wqkv->N / 32; out->hh = hd / 32; out->sink = sink; out->delay = 0;
    return MI355_OK;
}

extern "C" int mi355_linear_residual(const void* x, int32_t M, const mi355_weight_t* w, const void* bias, const void* residual_in,
                                     void* residual_out, float* tile_sumsq_out, int32_t tile_sumsq_ld, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x && residual_in && residual_out && M > 0, "linear_residual: bad args (M=%d)", M);
    MI355_CHECK_ARG(!tile_sumsq_out || (tile_sumsq_ld >= w->N / 16 && tile_sumsq_ld % 4 == 0), "linear_residual: tile_sumsq_ld=%d (>= N/16 = %d, multiple of 4)", tile_sumsq_ld, w->N / 16);
    if (M > 64 || !mi355_fullk_weight_ok(w)) return MI355_ERR_UNSUPPORTED;
    GemmParams p; fill_params(p, x, M, w);
    p.mode = MODE_F16; p.bias = (const f16*)bias; p.ldy = w->N;
    return mi355_gemm_fullk_residual(&p, w->wbits, w->group_size, residual_in, residual_out, tile_sumsq_out, tile_sumsq_ld, stream);
}

static int check_fused_norm(const mi355_fused_norm_t* n, int M, int K, const char* what) {
    MI355_CHECK_ARG(n->tile_sumsq && n->weight && n->tiles == K / 16 && n->tiles % 4 == 0 && n->ld >= n->tiles && n->ld % 4 == 0 && n->eps > 0.f,
                    "%s: fused norm needs tile_sumsq [M = %d][ld >= K/16 = %d, multiple of 4], weight and eps (got tiles=%d ld=%d)", what, M, K / 16, n->tiles, n->ld);
    return MI355_OK;
}

extern "C" int mi355_norm_linear(const void* h, int32_t M, const mi355_fused_norm_t* norm, const mi355_weight_t* w, const void* bias,
                                 void* y, int32_t epilogue, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(h && y && norm && M > 0, "norm_linear: bad args (M=%d)", M);
    if (int e = check_fused_norm(norm, M, w->K, "norm_linear")) return e;
    if (M > 16 || !mi355_fullk_weight_ok(w)) return MI355_ERR_UNSUPPORTED;
    GemmParams p; fill_params(p, h, M, w);
    p.mode = (epilogue & MI355_EPI_OUT_F32) ? MODE_F32 : (epilogue & MI355_EPI_SILU_MUL) ? MODE_SILU : MODE_F16;
    p.bias = (const f16*)bias; p.y = y; p.ldy = (p.mode == MODE_SILU) ? w->N / 2 : w->N;
    return mi355_gemm_fullk(&p, w->wbits, w->group_size, norm, stream);
}

extern "C" int mi355_qkv_rope_kv_write(const void* x, int32_t M, const mi355_weight_t* wqkv, const void* qkv_bias,
                                       const mi355_fused_norm_t* norm, const float* cos_sin, int32_t rope_dim, int32_t max_pos,
                                       const int32_t* positions, const int32_t* block_table, int32_t max_blocks_per_seq,
                                       int32_t q_len, int32_t nh, const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count,
                                       mi355_stream_t stream) {
    if (int e = check_weight(wqkv)) return e;
    MI355_CHECK_ARG(x && M > 0 && q_len >= 1 && M % q_len == 0, "qkv_rope_kv_write: M=%d q_len=%d", M, q_len);
    MI355_CHECK_ARG(kv && kv->kv_base && cos_sin && positions && block_table && q_out, "qkv_rope_kv_write: null pointer");
    MI355_CHECK_ARG(kv->page > 0 && nh > 0 && kv->nkv > 0 && max_pos > 0 && max_blocks_per_seq > 0 && kv->num_blocks > 0,
                    "qkv_rope_kv_write: bad dims");
    if (norm) if (int e = check_fused_norm(norm, M, wqkv->K, "qkv_rope_kv_write")) return e;
    if (M > (norm ? 16 : 64) || !mi355_fullk_weight_ok(wqkv) || rope_dim != kv->hd) return MI355_ERR_UNSUPPORTED;
    GemmParams p; fill_params(p, x, M, wqkv);
    p.mode = MODE_F16; p.bias = (const f16*)qkv_bias; p.ldy = wqkv->N;
    return mi355_gemm_fullk_rope(&p, wqkv->wbits, wqkv->group_size, norm, cos_sin, max_pos, positions, block_table, max_blocks_per_seq,
                                 q_len, nh, kv, q_out, oob_count, stream);
}

// ------------------------------------------------------------------ the same two launches for 1-64 rows (the step driver: from 5), activations as an image
// (gemm_fullk64.hip).  x_img: mi355_act_image_* of the [M][K] activations -- written directly by mi355_add_rmsnorm_img /
// mi355_paged_attn_rows_img, or by mi355_act_image_pack from a row-major tensor.
extern "C" int mi355_linear_residual_img(const void* x_img, int32_t M, const mi355_weight_t* w, const void* bias, const void* residual_in,
                                         void* residual_out, float* tile_sumsq_out, int32_t tile_sumsq_ld, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x_img && residual_in && residual_out && M > 0, "linear_residual_img: bad args (M=%d)", M);
    MI355_CHECK_ARG(!tile_sumsq_out || (tile_sumsq_ld >= w->N / 16 && tile_sumsq_ld % 4 == 0), "linear_residual_img: tile_sumsq_ld=%d (>= N/16 = %d, multiple of 4)", tile_sumsq_ld, w->N / 16);
    if (M < 1 || M > 64 || !img_weight_ok(w)) return MI355_ERR_UNSUPPORTED;
    GemmParams p; fill_params(p, x_img, M, w);
    p.mode = MODE_F16; p.bias = (const f16*)bias; p.ldy = w->N;
    return mi355_gemm_fullk_residual_img(&p, w->wbits, w->group_size, residual_in, residual_out, tile_sumsq_out, tile_sumsq_ld, nullptr, 0.f, nullptr, stream);
}

// Tensor parallelism (round 6): the row-parallel shard of a 1-64-row step -- O behind the attention image, down behind the SiLU image -- as ONE full-K
// launch whose epilogue writes y = 16-bit(xW + bias) into this rank's REGISTERED all-reduce buffer, where the next mi355_allreduce_fused_published_dt of
// the context pulls it from: no split-K slabs, no fold + publish stage in front of the flag exchange.  bias: pass it on rank 0 only (one bias for the sum).
// MI355_ERR_UNSUPPORTED (shape, format, or a context on the granule protocol): the caller stays on mi355_linear_partial* + mi355_allreduce_fused*.
// Reference slots: the row-parallel linears + all_reduce of modules/hybrid/causal_attention.py:91-92 and dense_mlp.py:104-105.
extern "C" int mi355_fullk64_publish_ok(const mi355_weight_t* w) {
    if (!img_weight_ok(w) || w->N % 32 != 0) return 0;
    if (!((w->wbits == 4 && w->group_size == 128) || (w->wbits == 8 && w->group_size == 0))) return 0;
    const int KC = w->K_pad / 128;
    if (KC <= (w->wbits == 8 ? 30 : 45)) return 1;
    return w->wbits == 4 && KC <= 75 && 2 * (w->N / 32) <= 256;   // five-chunk slices: <= 32 rows per block, i.e. the row-split form (gemm_fullk64.hip)
}
extern "C" int mi355_gemm_fullk_publish_img(const void* gp, int wbits, int group_size, const void* tgt, mi355_stream_t stream);
extern "C" int mi355_linear_publish_img(const void* x_img, int32_t M, const mi355_weight_t* w, const void* bias, mi355_allreduce_t* ar, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x_img && ar && M > 0, "linear_publish_img: bad args (M=%d)", M);
    if (M > 64 || !mi355_fullk64_publish_ok(w)) return MI355_ERR_UNSUPPORTED;
    mi355_publish_target_t tgt;
    if (int e = mi355_allreduce_publish_target(ar, M, w->N, &tgt)) return e;
    GemmParams p; fill_params(p, x_img, M, w);
    p.mode = MODE_F16; p.bias = (const f16*)bias; p.ldy = w->N;
    return mi355_gemm_fullk_publish_img(&p, w->wbits, w->group_size, &tgt, stream);
}

extern "C" int mi355_linear_residual_prenorm_img(const void* x_img, int32_t M, const mi355_weight_t* w, const void* bias, const void* residual_in,
                                                 void* residual_out, const void* norm_weight, int32_t norm_exp, void* xg_img_out,
                                                 float* tile_sumsq_out, int32_t tile_sumsq_ld, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x_img && residual_in && residual_out && norm_weight && xg_img_out && tile_sumsq_out && M > 0, "linear_residual_prenorm_img: bad args (M=%d)", M);
    MI355_CHECK_ARG(tile_sumsq_ld >= w->N / 16 && tile_sumsq_ld % 4 == 0 && w->N % 32 == 0 && norm_exp >= 0 && norm_exp <= 14,
                    "linear_residual_prenorm_img: tile_sumsq_ld=%d (>= N/16 = %d, multiple of 4), N %% 32, norm_exp=%d (0..14)", tile_sumsq_ld, w->N / 16, norm_exp);
    if (M < 1 || M > 64 || !img_weight_ok(w)) return MI355_ERR_UNSUPPORTED;
    GemmParams p; fill_params(p, x_img, M, w);
    p.mode = MODE_F16; p.bias = (const f16*)bias; p.ldy = w->N;
    return mi355_gemm_fullk_residual_img(&p, w->wbits, w->group_size, residual_in, residual_out, tile_sumsq_out, tile_sumsq_ld, norm_weight,
                                         ldexpf(1.f, -norm_exp), xg_img_out, stream);
}

extern "C" int mi355_linear_deferred_norm_img(const void* xg_img, int32_t M, const mi355_deferred_norm_t* dn, const mi355_weight_t* w,
                                              const void* bias, void* y, int32_t epilogue, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(xg_img && y && M > 0, "linear_deferred_norm_img: bad args (M=%d)", M);
    MI355_CHECK_ARG(!dn || (dn->tile_sumsq && dn->tiles == w->K / 16 && dn->tiles <= 512 && dn->tiles % 4 == 0 && dn->ld >= dn->tiles && dn->ld % 4 == 0 && dn->eps > 0.f && dn->unscale > 0.f),
                    "linear_deferred_norm_img: needs tile_sumsq [M][ld >= K/16 = %d, multiple of 4], eps and unscale", w->K / 16);
    if (M < 1 || M > 64 || !img_weight_ok(w)) return MI355_ERR_UNSUPPORTED;
    // bf16 tensors around the GEMM: taken when the output is an image again (fp16 of the bf16-rounded values) and there is no bias
    if (w->act_dtype == MI355_ACT_BF16 && (!(epilogue & MI355_EPI_OUT_IMAGE) || bias)) return MI355_ERR_UNSUPPORTED;
    GemmParams p; fill_params(p, xg_img, M, w);
    p.mode = (epilogue & MI355_EPI_OUT_F32) ? MODE_F32 : (epilogue & MI355_EPI_SILU_MUL) ? MODE_SILU : MODE_F16;
    p.bias = (const f16*)bias; p.y = y; p.ldy = (p.mode == MODE_SILU) ? w->N / 2 : w->N;
    p.x_img = 1; p.x_bytes = (uint32_t)mi355_act_image_bytes(M, w->K);
    if (epilogue & MI355_EPI_OUT_IMAGE) {
        MI355_CHECK_ARG(p.mode != MODE_F32 && p.ldy % 32 == 0, "linear_deferred_norm_img: an image output is a 16-bit tensor with a multiple of 32 columns");
        p.y_img = 1;
    }
    return mi355_gemm_wide_img(&p, w->wbits, w->group_size, dn, stream);
}

// y = epilogue(x W + bias) in ONE launch from an activation image, for a linear whose N does not fill the chip in the wide GEMM's form: a column-parallel
// shard under tensor parallelism (gemm_splitk64.hip, direct form).  MI355_ERR_UNSUPPORTED when the shape has no plan (the caller stays on mi355_linear_direct).
extern "C" int mi355_linear_direct_img(const void* x_img, int32_t M, const mi355_weight_t* w, const void* bias, void* y, int32_t epilogue,
                                       mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x_img && y && M > 0, "linear_direct_img: bad args (M=%d)", M);
    if (M < 1 || M > 64 || !img_weight_ok(w)) return MI355_ERR_UNSUPPORTED;
    if (w->act_dtype == MI355_ACT_BF16 && (!(epilogue & MI355_EPI_OUT_IMAGE) || bias)) return MI355_ERR_UNSUPPORTED;   // as mi355_linear_deferred_norm_img
    GemmParams p; fill_params(p, x_img, M, w);
    p.mode = (epilogue & MI355_EPI_OUT_F32) ? MODE_F32 : (epilogue & MI355_EPI_SILU_MUL) ? MODE_SILU : MODE_F16;
    p.bias = (const f16*)bias; p.y = y; p.ldy = (p.mode == MODE_SILU) ? w->N / 2 : w->N;
    p.x_img = 1; p.x_bytes = (uint32_t)mi355_act_image_bytes(M, w->K);
    if (epilogue & MI355_EPI_OUT_IMAGE) {
        MI355_CHECK_ARG(p.mode != MODE_F32 && p.ldy % 32 == 0, "linear_direct_img: an image output is a 16-bit tensor with a multiple of 32 columns");
        p.y_img = 1;
    }
    return mi355_gemm_splitk64_direct(&p, w->wbits, w->group_size, stream);
}

// Split-K slabs of a deep-K linear (down_proj) at 1-64 rows from an activation image (gemm_splitk64.hip): returns the number of
// fp32 slabs [n][M][N_pad] written to `partials` (to be folded by mi355_add_rmsnorm / _img), or MI355_ERR_UNSUPPORTED
extern "C" int mi355_linear_partial_img(const void* x_img, int32_t M, const mi355_weight_t* w, float* partials, int32_t max_splits,
                                        mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x_img && partials && M > 0 && max_splits >= 1, "linear_partial_img: bad args (M=%d)", M);
    if (M < 1 || M > 64 || !img_weight_ok(w)) return MI355_ERR_UNSUPPORTED;
    GemmParams p; fill_params(p, x_img, M, w);
    p.x_img = 1; p.x_bytes = (uint32_t)mi355_act_image_bytes(M, w->K); p.partials = partials;
    return mi355_gemm_splitk64(&p, w->wbits, w->group_size, max_splits > 16 ? 16 : max_splits, stream);
}

extern "C" int mi355_qkv_rope_kv_write_img(const void* x_img, int32_t M, const mi355_weight_t* wqkv, const void* qkv_bias,
                                           const float* cos_sin, int32_t rope_dim, int32_t max_pos, const int32_t* positions,
                                           const int32_t* block_table, int32_t max_blocks_per_seq, int32_t q_len, int32_t nh,
                                           const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count, mi355_stream_t stream) {
    if (int e = check_weight(wqkv)) return e;
    MI355_CHECK_ARG(x_img && M > 0 && q_len >= 1 && M % q_len == 0, "qkv_rope_kv_write_img: M=%d q_len=%d", M, q_len);
    MI355_CHECK_ARG(kv && kv->kv_base && cos_sin && positions && block_table && q_out, "qkv_rope_kv_write_img: null pointer");
    MI355_CHECK_ARG(kv->page > 0 && nh > 0 && kv->nkv > 0 && max_pos > 0 && max_blocks_per_seq > 0 && kv->num_blocks > 0,
                    "qkv_rope_kv_write_img: bad dims");
    if (M < 1 || M > 64 || !img_weight_ok(wqkv) || rope_dim != kv->hd) return MI355_ERR_UNSUPPORTED;
    GemmParams p; fill_params(p, x_img, M, wqkv);
    p.mode = MODE_F16; p.bias = (const f16*)qkv_bias; p.ldy = wqkv->N;
    return mi355_gemm_fullk_rope_img(&p, wqkv->wbits, wqkv->group_size, cos_sin, max_pos, positions, block_table, max_blocks_per_seq,
                                     q_len, nh, kv, q_out, oob_count, stream);
}
