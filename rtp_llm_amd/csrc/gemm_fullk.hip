// Weight-only W4 (and fp16-weight) GEMM with the WHOLE reduction inside one block ("full K"), gfx950: no split-K slabs, no reduce launch,
// the consumer's elementwise work fused into the epilogue.
//
// The decode step is a chain of short dependent launches; each one pays a launch + ramp + drain of ~5 us around
// ~1-10 us of streaming (b = 1: 2.1 ms for 4.5 GB).  The split-K linears made that worse twice over: fp32 slabs through
// memory (qkv / o / down: 1.9-2.5x the weight bytes, profiles/r01_pmc_hbm_traffic.txt) and a second launch to fold them.
// This kernel gives one block ONE or TWO 16-column tiles and all of K: its waves are the K slices (wave w owns chunks
// w, w + NW, w + 2 NW, ...), activations go straight from L2 into MFMA B-fragments (every x element is used by exactly
// one wave of the block, so LDS staging would buy nothing), the slices meet in LDS, and the summing waves run the
// epilogue:
//   FK_PLAIN  y = xW (+ bias), fp16 / fp32 / SiLU-mul store             (LinearBase.forward, linear_base.py:75-85)
//   FK_RESID  h' = h + fp16(xW + bias)                                  (the residual add after o_proj / down_proj of the
//                                                                        reference decoder layer; the fp32 slab sum +
//                                                                        add of add_rmsnorm_kernel without the slabs)
//   FK_ROPE   bias + NeoX RoPE + Q extract + paged fp16 KV write        (FusedRopeKVCacheDecodeOp::forward,
//             for the tile pair (d, d + hd/2) of one head                FusedRopeKVCacheOp.cc:519-646; same arithmetic
//                                                                        as rope_kv.hip, which stays for INT8 caches)
// Dequant is the operand-side sequence of gemm.hip (exact subtract of the biased code, one rounding, 13 VALU per 8
// weights) for every M: at M <= 16 the kernel is latency-, not issue-bound.
#include "gemm_fullk.h"

namespace {


// XL: activation wave-loads per chunk at <= 16 rows.  A B fragment of the 16x16x32 MFMA is 16 rows x 32 k; with M rows alive
// a load in that shape is M / 16 dense, and at a few rows the launches were bound by the number of such loads (four per
// chunk and as many again for gamma, against ONE 1 KB weight load per tile: profiles/r03_fullk_fixed_costs.txt).  XL = 1
// (M <= 4): lane (jj, q) loads row jj % 4 at k-step jj / 4 -- one load is the whole chunk of four rows; XL = 2 (M <= 8): row
// jj % 8, k-step 2 l + jj / 8.  Step s is then a DPP row shift of the register (lane jj < R takes lane jj + R s', R = 4 XL rows per load); lanes
// >= R hold other steps' data, i.e. garbage in accumulator rows >= R that are never stored.  XL = 4: the fragment shape.
//
// HELP (<= 16 rows with a fused norm or the RoPE epilogue): the block gets one more wave that owns no K slice.  It adds up
// the norm's partial sums and publishes 1 / rms through LDS (the K waves poll for it after their requests are out: no barrier
// between a wave's requests and its first MFMA), walks the position -> block id -> rotation row chain, and runs the epilogue
// after the slices met.  Both used to sit in front of wave 0's weight requests with the whole block waiting behind it.
template <int GS, int MB, int TPB, int EPI, bool NORM, int XL = 4>
__global__ __launch_bounds__(MB >= 3 ? (TPB >= 2 ? 512 : 768) : 1024) void gemm_fullk_kernel(const FullKParams fp) {
    constexpr bool W16 = GS == 0;                    // GS: 4 -> W4 g128, 2 -> g64, 1 -> g32, 0 -> fp16 weights (no meta, 4 wave-loads per chunk)
    constexpr int LPC = W16 ? 4 : 1;
    constexpr int NSUB = W16 ? 1 : 4 / (W16 ? 1 : GS), SPG = 4 / NSUB;
    constexpr int WD = (GS == 0 && TPB >= 2) ? 1 : 2; // weight ring, chunks (fp16 tiles are 4x the registers).  Deeper rings for down_proj's ten-chunk
                                                      // slices do not pay (3 / 4 / 5 chunks: 13.6 / 13.0 / 11.5 us against 11.3): its loop already streams at ~6 TB/s
    constexpr int HD = ((MB >= 3 && TPB < 2) || (MB == 2 && TPB >= 2) || NORM || TPB >= 4 || WD == 1) ? 2 : 4;   // activation ring, half chunks (2 k-steps x MB row blocks each); by register budget
    constexpr bool HELP = MB == 1 && (NORM || EPI == FK_ROPE);
    constexpr int RPL = 4 * XL, SPL = 4 / XL;        // XL < 4: rows per activation load, k-steps per load
    static_assert(XL == 4 || (MB == 1 && !W16 && WD == 2), "dense activation loads: <= 8 rows, W4");
    static_assert(!NORM || MB == 1, "the on-the-fly norm is built for <= 16 rows");
    constexpr uint32_t FLAGS = 0x00020000u, OOBX = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);     // [NW][TPB * MB][64]
    const GemmParams& p = fp.g;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW   = (int)(blockDim.x >> 6) - (HELP ? 1 : 0);   // K-slice waves
    const bool helper = HELP && wave == NW;
    const int jj = lane & 15, q = lane >> 4;
    // NORM: 1 / rms of rows 0..15, 0 = not there yet.  An LDS-qualified pointer: a volatile access through a generic one compiles to
    // flat_load / flat_store, which count on vmcnt -- the K waves' poll would then wait for every weight request in flight
    typedef __attribute__((address_space(3))) volatile float lds_vfloat;
    lds_vfloat* rs_sh = (lds_vfloat*)(smem + (size_t)NW * TPB * MB * 1024);
    FK_STAMP(0);
    // ---- the helper's first job: 1 / rms of every row from the producer's per-tile partial sums.  The sums of the first rows (all of
    // them at <= 8 rows) are requested at once, and BEFORE the K waves are let go (the barrier that also covers the zeroing of
    // rs_sh): behind their burst of weight requests these few hundred bytes queued for microseconds, and no K wave can start
    // its first MFMA without them.  Register budget: NRA rows x 4 (the K waves' setup is hoisted above the role branch).
    constexpr int NR  = NORM ? (XL == 1 ? 4 : XL == 2 ? 8 : 16) : 1;   // rows the instance can see
    constexpr int NRA = NR < 8 ? NR : 8;                                // rows of the first batch
    u32x4 hsq[NRA];
    if constexpr (NORM) {
        if (helper) {
            __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)fp.ssq_in, 0, (uint32_t)p.M * fp.ssq_ld * 4u, FLAGS);
            const int nv = fp.ssq_tiles >> 2;
#pragma unroll
            for (int r = 0; r < NRA; ++r)
                hsq[r] = bload128<0>(rq, (r < p.M && lane < nv) ? (uint32_t)(r * fp.ssq_ld + lane * 4) * 4u : OOBX);
        }
        if (threadIdx.x < 16) rs_sh[threadIdx.x] = 0.f;
        // the LDS stores above complete, then the barrier, with the helper's requests still in flight across it (spelled out: this is
        // what __syncthreads() compiles to on gfx950 -- lgkmcnt(0) + s_barrier, no vmcnt drain -- and nothing weaker or stronger will do)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    int tile[TPB];
    if constexpr (EPI == FK_ROPE) {
        const int hh = fp.r.hd >> 5;                 // tiles per half head
        const int h = blockIdx.x / hh, j = blockIdx.x % hh;
        tile[0] = h * 2 * hh + j;
        tile[TPB - 1] = tile[0] + hh;
    } else {
#pragma unroll
        for (int t = 0; t < TPB; ++t) tile[t] = blockIdx.x * TPB + t;
    }

    auto publish_rs = [&]() {                          // helper: reduce and publish (after the independent epilogue requests are out)
        if constexpr (NORM) {
            __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)fp.ssq_in, 0, (uint32_t)p.M * fp.ssq_ld * 4u, FLAGS);
            const int nv = fp.ssq_tiles >> 2;
            auto row_sum = [&](int r, const u32x4& first) {   // lanes' shares of row r -> 1 / rms in LDS
                const f32x4 t = __builtin_bit_cast(f32x4, first);
                float a = (t[0] + t[1]) + (t[2] + t[3]);
                for (int v0 = 64; v0 < nv; v0 += 64) {         // K > 4096: the columns past the first pass
                    const f32x4 u = __builtin_bit_cast(f32x4, bload128<0>(rq, v0 + lane < nv ? (uint32_t)(r * fp.ssq_ld + (v0 + lane) * 4) * 4u : OOBX));
                    a += (u[0] + u[1]) + (u[2] + u[3]);
                }
                // 16-lane rows by DPP, the four rows by permlane swaps: VALU only, the reductions of the rows pipeline
                a = dpp_add<0xB1>(a); a = dpp_add<0x4E>(a); a = dpp_add<0x141>(a); a = dpp_add<0x140>(a);   // quad xor 1, xor 2, row_half_mirror, row_mirror
                a = xor16_sum(a); a = xor32_sum(a);
                if (lane == 0) rs_sh[r] = rsqrtf(a / (float)p.K + fp.eps);
            };
#pragma unroll
            for (int r = 0; r < NRA; ++r)
                if (r < p.M) row_sum(r, hsq[r]);
            if constexpr (NR > NRA) {                  // rows 8..15 (9 to 16 rows only): a second round trip
                if (p.M > NRA) {
                    u32x4 more[NR - NRA];
#pragma unroll
                    for (int r = NRA; r < NR; ++r)
                        more[r - NRA] = bload128<0>(rq, (r < p.M && lane < nv) ? (uint32_t)(r * fp.ssq_ld + lane * 4) * 4u : OOBX);
#pragma unroll
                    for (int r = NRA; r < NR; ++r)
                        if (r < p.M) row_sum(r, more[r - NRA]);
                }
            }
        }
    };

    // ---- the epilogue's wave(s) (the helper; without one, wave mb < MB for row block mb) request their operands now: residual /
    // bias rows, position -> block id -> rotation row would otherwise be two to three dependent round trips after the last barrier
    const int  mb_epi = HELP ? 0 : wave;
    const int  m_epi = mb_epi * 16 + jj;              // row of this lane in the epilogue
    const bool epi_wave = (HELP ? helper : wave < MB) && m_epi < p.M;
    f16x4 pre_res[TPB], pre_bias[TPB];
    int   pre_pos = 0, pre_blk = 0;
    f32x4 pre_cs[2];
#pragma unroll
    for (int t = 0; t < TPB; ++t) { pre_res[t] = (f16x4){0, 0, 0, 0}; pre_bias[t] = pre_res[t]; }
    pre_cs[0] = pre_cs[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (epi_wave) {
        if constexpr (EPI == FK_RESID) {
#pragma unroll
            for (int t = 0; t < TPB; ++t) {
                const int n0 = tile[t] * 16 + q * 4;
                if (n0 < p.N) {
                    pre_res[t] = *reinterpret_cast<const f16x4*>(fp.res_in + (size_t)m_epi * p.N + n0);
                    if (p.bias) pre_bias[t] = *reinterpret_cast<const f16x4*>(p.bias + n0);
                }
            }
        } else if constexpr (EPI == FK_ROPE) {
            const RopeEpi& Rp = fp.r;
            const int half = Rp.hd >> 1, hh = Rp.hd >> 5;
            const int h = tile[0] / (2 * hh), d0 = (tile[0] % (2 * hh)) * 16 + q * 4;
            if (p.bias) {
                pre_bias[0] = *reinterpret_cast<const f16x4*>(p.bias + h * Rp.hd + d0);
                pre_bias[TPB - 1] = *reinterpret_cast<const f16x4*>(p.bias + h * Rp.hd + d0 + half);
            }
            pre_pos = Rp.positions[m_epi];
        }
    }
    if (helper) publish_rs();
    if (epi_wave) {
        if constexpr (EPI == FK_ROPE) {              // the dependent part of the chain
            const RopeEpi& Rp = fp.r;
            const int half = Rp.hd >> 1, hh = Rp.hd >> 5;
            const int d0 = (tile[0] % (2 * hh)) * 16 + q * 4;
            const int pos = min(max(pre_pos, 0), min(Rp.max_pos, Rp.max_blocks * Rp.page) - 1);
            pre_blk = Rp.block_table[(size_t)(m_epi / Rp.q_len) * Rp.max_blocks + pos / Rp.page];
            const float* cs = Rp.cos_sin + ((size_t)pos * half + d0) * 2;
            pre_cs[0] = *reinterpret_cast<const f32x4*>(cs); pre_cs[1] = *reinterpret_cast<const f32x4*>(cs + 4);
        }
    }

    if (helper) {
        FK_STAMP(1);
    } else {
    // =================================================================== K-slice waves
    // K slices are interleaved (fp.ilv): wave w owns chunks w, w + NW, w + 2 NW, ... -- at any moment the block's outstanding
    // requests cover one dense run of each tile's weights; otherwise wave w owns the run [w KC / NW, (w + 1) KC / NW)
    const int c0 = fp.ilv ? wave : (wave * p.KC) / NW, cs = fp.ilv ? NW : 1;
    const int n_ch = fp.ilv ? (p.KC - wave + NW - 1) / NW : ((wave + 1) * p.KC) / NW - c0;
    const int span = fp.ilv ? p.KC - c0 : n_ch;      // chunks from the wave's first to the end of what its descriptors cover

    __amdgpu_buffer_rsrc_t rw[TPB], rm[TPB];
#pragma unroll
    for (int t = 0; t < TPB; ++t) {
        const bool ok = tile[t] < p.NT;
        // descriptors start at the wave's first chunk and end with its last one (interleaved: with the tile): a slot past the wave's share is past the end
        const char* wb = (const char*)p.qw + ((size_t)tile[t] * p.KC + c0) * (LPC * 1024);
        rw[t] = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (ok && n_ch > 0) ? span * LPC * 1024 : 0, FLAGS);
        const char* mb = (const char*)p.meta + ((size_t)c0 * NSUB * p.N_pad + tile[t] * 16) * 4;
        rm[t] = __builtin_amdgcn_make_buffer_rsrc((void*)(W16 ? (const char*)p.qw : mb), 0, (ok && n_ch > 0 && !W16) ? ((span * NSUB - 1) * p.N_pad + 16) * 4 : 0, FLAGS);
    }
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, FLAGS);

    // B fragment (mb, chunk i, k-step s): row 16 mb + jj, k = 128 (c0 + i cs) + 32 s + 8 q .. + 7
    uint32_t xv[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) xv[mb] = (mb * 16 + jj < p.M) ? (uint32_t)(((size_t)(mb * 16 + jj) * p.K + c0 * 128 + q * 8) * 2) : OOBX;
    // XL < 4: this lane's share of a dense load: row jj % RPL at k-step (load l) SPL + jj / RPL
    const int rowd = jj % RPL, subd = jj / RPL;
    const uint32_t xvd = (rowd < p.M) ? (uint32_t)(((size_t)rowd * p.K + c0 * 128 + subd * 32 + q * 8) * 2) : OOBX;
    int nhc = 2 * n_ch;                              // half chunks of this wave; a VGPR so that the range select below is a
    asm volatile("" : "+v"(nhc));                    // v_cndmask, not a branch around the loads (which would drain vmcnt)

    float rs[MB];
    __amdgpu_buffer_rsrc_t rg = rx;
    if constexpr (NORM) rg = __builtin_amdgcn_make_buffer_rsrc((void*)fp.gamma, 0, p.K * 2, FLAGS);
    const uint32_t lane16 = lane * 16u, jj4 = jj * 4u;
    const uint32_t gv = (uint32_t)((c0 * 128 + q * 8) * 2), gvd = (uint32_t)((c0 * 128 + subd * 32 + q * 8) * 2);
    u32x4    gr[(NORM && XL == 4) ? HD : 1][2];      // gamma of the half chunk's two k-steps (k = the fragment's 8 columns)
    u32x4    wr[WD][TPB][LPC];
    uint32_t mr[WD][TPB][NSUB];
    // MM (W4, several tiles per block): the (zero, scale) words of up to four tiles come in ONE load -- lane (jj, q) asks for column
    // jj of tile q -- and tile t's word is then read off lane (jj, t) by ds_bpermute (the crossbar, no LDS memory): the launch
    // is bound by the number of vector-memory instructions, not by their bytes.  A fifth tile keeps its own load.
    constexpr bool MM = !W16 && TPB >= 2;
    constexpr int  MT = TPB < 4 ? TPB : 4;           // tiles of the merged load
    uint32_t mq[MM ? WD : 1][NSUB];
    int tq = tile[0];
#pragma unroll
    for (int t = 1; t < MT; ++t) tq = (q == t) ? tile[t] : tq;
    const uint32_t mqv = (MM && q < MT && tq < p.NT) ? (uint32_t)(tq * 16 + jj) * 4u : OOBX;
    __amdgpu_buffer_rsrc_t rmq = rx;
    if constexpr (MM)
        rmq = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.meta + (size_t)c0 * NSUB * p.N_pad * 4), 0, n_ch > 0 ? span * NSUB * p.N_pad * 4 : 0, FLAGS);
    u32x4    xr[XL == 4 ? HD : 1][MB][2];
    u32x4    xq[XL < 4 ? WD : 1][XL < 4 ? XL : 1], gq[(NORM && XL < 4) ? WD : 1][XL < 4 ? XL : 1];   // dense loads: [chunk slot][load]
    f32x4    acc[TPB][MB];
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_w = [&](int d, int i) {                // past the wave's range: out of the descriptor, returns 0, no traffic
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
#pragma unroll
            for (int lp = 0; lp < LPC; ++lp) wr[d][t][lp] = bload128<2 /*nt*/>(rw[t], lane16, (uint32_t)(i * cs * LPC + lp) * 1024u);
            if constexpr (!W16) {
#pragma unroll
                for (int gi = 0; gi < NSUB; ++gi)
                    if (!MM || t >= MT) mr[d][t][gi] = __builtin_amdgcn_raw_buffer_load_b32(rm[t], jj4, (uint32_t)(i * cs * NSUB + gi) * (uint32_t)p.N_pad * 4u, 0);
            }
        }
        if constexpr (MM) {
#pragma unroll
            for (int gi = 0; gi < NSUB; ++gi) mq[d][gi] = __builtin_amdgcn_raw_buffer_load_b32(rmq, mqv, (uint32_t)(i * cs * NSUB + gi) * (uint32_t)p.N_pad * 4u, 0);
        }
    };
    auto spread_meta = [&](int d) {                  // MM: tile t's words from lane (jj, t), once per ring chunk
        if constexpr (MM) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int gi = 0; gi < NSUB; ++gi) mr[d][t][gi] = (uint32_t)__builtin_amdgcn_ds_bpermute((t * 16 + jj) * 4, (int)mq[d][gi]);
        }
    };
    auto load_x = [&](int d, int hc) {               // past the range the next wave's slice would be read: force zeros
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) xr[d][mb][ss] = bload128<0>(rx, hc < nhc ? xv[mb] : OOBX, (uint32_t)((hc >> 1) * cs) * 256u + (hc & 1) * 128u + ss * 64u);
        if constexpr (NORM) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) gr[d][ss] = bload128<0>(rg, hc < nhc ? gv : OOBX, (uint32_t)((hc >> 1) * cs) * 256u + (hc & 1) * 128u + ss * 64u);
        }
    };
    auto load_xd = [&](int d, int i) {               // XL < 4: chunk slot i of this wave, XL dense loads (+ gamma in the same shape)
#pragma unroll
        for (int l = 0; l < (XL < 4 ? XL : 1); ++l) {
            xq[d][l] = bload128<0>(rx, 2 * i < nhc ? xvd : OOBX, (uint32_t)(i * cs) * 256u + l * (SPL * 64));
            if constexpr (NORM) gq[d][l] = bload128<0>(rg, 2 * i < nhc ? gvd : OOBX, (uint32_t)(i * cs) * 256u + l * (SPL * 64));
        }
    };
    auto normed = [&](const u32x4& hraw, const u32x4& graw, float r) -> f16x8 {   // gamma * fp16(h * rs): add_rmsnorm_kernel's arithmetic
        const f16x8 h = __builtin_bit_cast(f16x8, hraw), g = __builtin_bit_cast(f16x8, graw);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = g[e] * (f16)((float)h[e] * r);
        return o;
    };
    const W4Consts w4c = w4_consts();
    const f16x2 c960 = {(f16)960.f, (f16)960.f};
    auto weights_of = [&](int d, int t, int s) -> f16x8 {   // A fragment of k-step s of ring chunk d, tile t
        if constexpr (W16) {
            return __builtin_bit_cast(f16x8, wr[d][t][s]);
        } else {
            const uint32_t m = mr[d][t][s / SPG];
            const f16x2 zn = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u)), sc = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
            return dequant_w4_vc(wr[d][t][0][s], zn, zn + c960, sc, w4c);
        }
    };
    auto half_chunk = [&](int d, int xb, int hf) {   // k-steps 2 hf, 2 hf + 1 of ring chunk d against activation buffer xb
        if (hf == 0) spread_meta(d);
        if constexpr (NORM) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) xr[xb][mb][ss] = __builtin_bit_cast(u32x4, normed(xr[xb][mb][ss], gr[xb][ss], rs[mb]));
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int s = 2 * hf + ss;
#pragma unroll
            for (int t = 0; t < TPB; ++t) {
                const f16x8 a = weights_of(d, t, s);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[t][mb] = mfma16x16x32(a, __builtin_bit_cast(f16x8, xr[xb][mb][ss]), acc[t][mb]);
            }
        }
    };
    auto chunk_dense = [&](int d) {                  // XL < 4: the four k-steps of ring chunk d against its dense activation loads
        spread_meta(d);
        if constexpr (NORM) {
#pragma unroll
            for (int l = 0; l < (XL < 4 ? XL : 1); ++l) xq[d][l] = __builtin_bit_cast(u32x4, normed(xq[d][l], gq[d][l], rs[0]));
        }
        static_for<0, 4>([&](auto s_) {
            constexpr int s = decltype(s_)::value, l = (XL < 4) ? s / SPL : 0, sb = (XL < 4) ? s % SPL : 0;
            u32x4 b = xq[d][l];
            if constexpr (sb > 0) {                  // lane jj < RPL takes lane jj + RPL sb of its 16-lane row
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xq[d][l][e], 0x100 + RPL * sb, 0xF, 0xF, true);
            }
#pragma unroll
            for (int t = 0; t < TPB; ++t) acc[t][0] = mfma16x16x32(weights_of(d, t, s), __builtin_bit_cast(f16x8, b), acc[t][0]);
        });
    };

    // ---- everything the first round needs is requested before the first wait
    if constexpr (XL < 4) {
#pragma unroll
        for (int d = 0; d < WD; ++d) load_xd(d, d);
    } else {
#pragma unroll
        for (int d = 0; d < HD; ++d) load_x(d, d);
    }
#pragma unroll
    for (int d = 0; d < WD; ++d) load_w(d, d);
    // NORM: the helper has had a round trip's head start; pick up 1 / rms of this lane's row(s) once it is there (bounded: a
    // row that never arrives would hang the device, so after ~1e6 polls the wave goes on with NaN and the output shows it)
    if constexpr (NORM) {
        const int myrow = XL < 4 ? rowd : jj;
        float r = 0.f;
        for (int spin = 0; spin < (1 << 20); ++spin) {
            r = myrow < p.M ? rs_sh[myrow] : 1.f;
            if (!__builtin_amdgcn_ballot_w64(r == 0.f)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        rs[0] = myrow < p.M ? (r == 0.f ? __builtin_nanf("") : r) : 0.f;
    }
    FK_STAMP(1);
    if constexpr (XL < 4) {
        for (int i = 0; i < n_ch; i += WD) {
#pragma unroll
            for (int c = 0; c < WD; ++c) {
                chunk_dense(c);
#ifdef MI355_FULLK_STAMPS
                if (i == 0 && c == 0) { asm volatile("s_nop 0" ::: "memory"); FK_STAMP(2); }
#endif
                load_xd(c, i + c + WD);
                load_w(c, i + c + WD);
            }
        }
    } else {
    // rounds of WD chunks = 2 WD half chunks, no guards inside (slots past the end multiply zero activations)
    for (int i = 0; i < n_ch; i += WD) {
#pragma unroll
        for (int u = 0; u < 2 * WD; ++u) {
            const int hc = 2 * i + u;
            half_chunk(u >> 1, u % HD, u & 1);
#ifdef MI355_FULLK_STAMPS
            if (i == 0 && u == 0) { asm volatile("s_nop 0" ::: "memory"); FK_STAMP(2); }
#endif
            load_x(u % HD, hc + HD);
            if (u & 1) load_w(u >> 1, i + (u >> 1) + WD);
        }
    }
    }

    FK_STAMP(3);
    // ---- the K slices meet in LDS
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) red[((size_t)wave * (TPB * MB) + t * MB + mb) * 64 + lane] = acc[t][mb];
    }   // K-slice waves
    __syncthreads();
    FK_STAMP(4);

    if constexpr (EPI == FK_PLAIN) {                 // wave e sums (in slice order) and stores (tile e / MB, row block e % MB)
        if (helper) return;
        for (int e = wave; e < TPB * MB; e += NW) {
            const int t = e / MB, mbe = e % MB, me = mbe * 16 + jj;
            f32x4 ve = {0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < NW; ++w) ve += red[((size_t)w * (TPB * MB) + t * MB + mbe) * 64 + lane];
            if (me < p.M && tile[t] < p.NT) gemm_store(p, ve, me, tile[t] * 16 + q * 4, 0);
        }
        FK_STAMP(5);
        return;
    }
    // the epilogue's wave sums row block mb_epi of every tile of the block, in slice order
    if (HELP ? !helper : wave >= MB) return;
    const int mb = mb_epi;
    f32x4 v[TPB];
#pragma unroll
    for (int t = 0; t < TPB; ++t) {
        v[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < NW; ++w) v[t] += red[((size_t)w * (TPB * MB) + t * MB + mb) * 64 + lane];
    }
    const int m = mb * 16 + jj;
    if (m >= p.M) return;

    if constexpr (EPI == FK_RESID) {
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
            const int n0 = tile[t] * 16 + q * 4;
            if (n0 >= p.N) continue;
            const f16x4 bv = pre_bias[t], rin = pre_res[t];
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = (float)(f16)(v[t][r] + (float)bv[r]);    // the linear's output is an fp16 tensor in the reference
                o[r] = (f16)(y + (float)rin[r]);
            }
            *reinterpret_cast<f16x4*>(fp.res_out + (size_t)m * p.N + n0) = o;
            if (fp.ssq_out) {                         // this tile's share of sum h'^2 of the row, for the consumer's RMSNorm
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) a += (float)o[r] * (float)o[r];
                a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                if (q == 0) fp.ssq_out[(size_t)m * fp.ssq_ld + tile[t]] = a;
            }
        }
        FK_STAMP(5);
    } else {
        // tile pair of one head: this lane holds dims d0..d0+3 (v[0]) and d0+half..+3 (v[1]) of row m = token m
        const RopeEpi& R = fp.r;
        const int half = R.hd >> 1, hh = R.hd >> 5;
        const int h  = tile[0] / (2 * hh);
        const int d0 = (tile[0] % (2 * hh)) * 16 + q * 4;
        float x0[4], x1[4];
        const f16x4 b0 = pre_bias[0], b1 = pre_bias[TPB - 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x0[r] = (float)(f16)(v[0][r] + (float)b0[r]);
            x1[r] = (float)(f16)(v[TPB - 1][r] + (float)b1[r]);
        }
        const int pos_in  = pre_pos;
        const int pos_lim = min(R.max_pos, R.max_blocks * R.page);
        const int pos = min(max(pos_in, 0), pos_lim - 1);
        const bool is_v = h >= R.nh + R.nkv;
        if (!is_v) {
            const f32x4 cs01 = pre_cs[0], cs23 = pre_cs[1];
            const float c[4] = {cs01[0], cs01[2], cs23[0], cs23[2]}, s[4] = {cs01[1], cs01[3], cs23[1], cs23[3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float r0 = c[r] * x0[r] - s[r] * x1[r];
                const float r1 = c[r] * x1[r] + s[r] * x0[r];
                x0[r] = (float)(f16)r0; x1[r] = (float)(f16)r1;
            }
        }
        f16x4 o0, o1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] = (f16)x0[r]; o1[r] = (f16)x1[r]; }
        FK_STAMP(5);
        if (h < R.nh) {
            f16* dst = R.q_out + ((size_t)m * R.nh + h) * R.hd + d0;
            *reinterpret_cast<f16x4*>(dst) = o0;
            *reinterpret_cast<f16x4*>(dst + half) = o1;
            return;
        }
        const int kh = is_v ? h - R.nh - R.nkv : h - R.nh;
        if (pos_in < 0) return;                                  // padding row of a multi-row step
        const int blk = pre_blk;
        if (pos != pos_in || blk < 0 || blk >= R.num_blocks) {   // stale position / block id: never write somebody else's page
            if (h == R.nh && d0 == 0 && R.oob_count) atomicAdd(R.oob_count, 1);
            return;
        }
        const int tok = pos % R.page;
        const size_t head_elems = (size_t)R.page * R.hd;
        const size_t blk_base   = ((size_t)blk * 2 + (is_v ? 1 : 0)) * R.nkv + kh;
        f16* dst = (f16*)R.kv_base + blk_base * head_elems;
        if (!is_v) {
            *reinterpret_cast<f16x4*>(dst + tok * R.hd + d0) = o0;
            *reinterpret_cast<f16x4*>(dst + tok * R.hd + d0 + half) = o1;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { dst[(d0 + r) * R.page + tok] = o0[r]; dst[(d0 + r + half) * R.page + tok] = o1[r]; }
        }
    }
}

template <int GS, int MB, int TPB, int EPI, bool NORM, int XL = 4>
int launch_fullk_t(const FullKParams& fp, int blocks, int NW, hipStream_t st) {
    auto k = gemm_fullk_kernel<GS, MB, TPB, EPI, NORM, XL>;
    constexpr int HELP = (MB == 1 && (NORM || EPI == FK_ROPE)) ? 1 : 0;   // the wave without a K slice (see the kernel)
    const size_t lds = (size_t)NW * TPB * MB * 1024 + (NORM ? 256 : 0);
    if (lds > 64 * 1024)
        if (int e = raise_dynamic_lds((const void*)k, "gemm_fullk")) return e;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * (NW + HELP)), lds, st, fp);
    MI355_CHECK_LAUNCH("gemm_fullk_kernel");
    return MI355_OK;
}

// waves per block = K slices (+ the helper), bounded by the register budget of the shape and by `wave_cap` (callers that want
// several blocks per CU: 32 wave slots)
template <int TPB, int EPI, bool NORM>
int launch_fullk(const FullKParams& fp, int group_size, int blocks, hipStream_t st, int wave_cap = 16) {
    const GemmParams& g = fp.g;
    const int MB = g.M <= 16 ? 1 : (g.M <= 32 ? 2 : 4);
    if (NORM && MB > 1) return MI355_ERR_UNSUPPORTED;       // the on-the-fly norm is instantiated for <= 16 rows (latency regime)
    int maxw = MB >= 3 ? (TPB >= 2 ? 8 : 12) : 16;
    if (maxw > wave_cap) maxw = wave_cap;
    if (MB == 1 && (NORM || EPI == FK_ROPE)) maxw -= 1;
    const int cpw = cdiv(g.KC, maxw), NW = cdiv(g.KC, cpw);
    if (NW < MB) return MI355_ERR_UNSUPPORTED;
    // dense activation loads at <= 8 rows (g128): one or two loads per chunk instead of four
    const int xl = (group_size == 128 && MB == 1 && TUNE(5) != 1) ? (g.M <= 4 ? 1 : g.M <= 8 ? 2 : 4) : 4;
#define FK_(GS_)                                                                                   \
    if constexpr (NORM || TPB >= 4) { if (MB > 1) return MI355_ERR_UNSUPPORTED; return launch_fullk_t<GS_, 1, TPB, EPI, NORM>(fp, blocks, NW, st); } \
    else switch (MB) {                                                                             \
    case 1: return launch_fullk_t<GS_, 1, TPB, EPI, false>(fp, blocks, NW, st);                    \
    case 2: return launch_fullk_t<GS_, 2, TPB, EPI, false>(fp, blocks, NW, st);                    \
    default: return launch_fullk_t<GS_, 4, TPB, EPI, false>(fp, blocks, NW, st);                   \
    }
    if (group_size == 128) {
        if (xl == 1) return launch_fullk_t<4, 1, TPB, EPI, NORM, 1>(fp, blocks, NW, st);
        if (xl == 2) return launch_fullk_t<4, 1, TPB, EPI, NORM, 2>(fp, blocks, NW, st);
        FK_(4)
    }
    if constexpr (TPB < 4) {     // the wide few-row shape exists for g128 only
        if (group_size == 64) { FK_(2) }
        if (group_size == 32) { FK_(1) }
        if (group_size == 0) { FK_(0) }
    }
#undef FK_
    return MI355_ERR_UNSUPPORTED;
}

// shapes this kernel takes: W4 group-wise or fp16 weights, whole chunks, x image below the OOB offset
bool fullk_shape_ok(const GemmParams& g, int wbits, int& group_size) {
    if (wbits == 16) group_size = 0;                 // fp16 weights: the GS = 0 instances
    return ((wbits == 4 && (group_size == 128 || group_size == 64 || group_size == 32)) || wbits == 16) && g.M >= 1 && g.M <= 64 && g.K % 128 == 0 &&
           g.K == g.KC * 128 && (uint64_t)g.M * g.K * 2 < 0x7FFFFFF0ull && g.KC >= 4;
}

} // namespace

// the launches on activation images (gemm_fullk64.hip): W4 group-wise or per-channel W8 (group_size 0)
static bool fullk64_shape_ok(const GemmParams& g, int wbits, int group_size) {
    const bool fmt = (wbits == 4 && (group_size == 128 || group_size == 64 || group_size == 32)) || (wbits == 8 && group_size == 0);
    return fmt && g.M >= 1 && g.M <= 64 && g.K % 128 == 0 && g.K == g.KC * 128 && (uint64_t)g.M * g.K * 2 < 0x7FFFFFF0ull && g.KC >= 4;
}

#ifdef MI355_FULLK_STAMPS
unsigned long long* g_fullk_stamps = nullptr;   // also read by gemm_fullk64.hip
extern "C" void mi355_debug_fullk_stamps(void* p) { g_fullk_stamps = (unsigned long long*)p; }
#define FK_SET_STAMPS(fp) (fp).stamps = g_fullk_stamps
#else
#define FK_SET_STAMPS(fp) do { } while (0)
#endif

// 1-64 rows, W4 group-wise (K <= 5760) or per-channel W8 (group_size 0, K <= 3840), activations as an image (mi355_act_image_*): gemm_fullk64.hip
extern "C" int mi355_gemm_fullk64(const void* fp, int epi, int group_size, mi355_stream_t stream);

static void set_norm(FullKParams& fp, const mi355_fused_norm_t* n) {
    fp.ssq_in = n->tile_sumsq; fp.ssq_tiles = n->tiles; fp.ssq_ld = n->ld; fp.gamma = (const f16*)n->weight; fp.eps = n->eps;
}

// y = [RMSNorm](x) W (+ bias) with the epilogue of p.mode (fp16 / fp32 / SiLU-mul), one launch, no workspace.
// norm != null: x is the un-normed row, see FullKParams.  Wide outputs at a few rows (gate_up) take five tiles per block and
// two blocks per CU so that one wave of blocks covers the matrix.
extern "C" int mi355_gemm_fullk(const void* gp, int wbits, int group_size, const void* norm, mi355_stream_t stream) {
    FullKParams fp{};
    fp.g = *reinterpret_cast<const GemmParams*>(gp);
    FK_SET_STAMPS(fp);
    fp.ilv = TUNE(4) ? TUNE(4) - 1 : 1;
    if (!fullk_shape_ok(fp.g, wbits, group_size) || fp.g.mode == MODE_PARTIAL) return MI355_ERR_UNSUPPORTED;
    const bool wide = fp.g.NT >= 1024 && fp.g.M <= 16 && group_size == 128;     // 5 tiles per block, 8 waves: two blocks per CU, one wave of blocks for gate_up
    if (norm) {
        set_norm(fp, (const mi355_fused_norm_t*)norm);
        return wide ? launch_fullk<5, FK_PLAIN, true>(fp, group_size, cdiv(fp.g.NT, 5), (hipStream_t)stream, 8)
                    : launch_fullk<1, FK_PLAIN, true>(fp, group_size, fp.g.NT, (hipStream_t)stream);
    }
    return wide ? launch_fullk<5, FK_PLAIN, false>(fp, group_size, cdiv(fp.g.NT, 5), (hipStream_t)stream, 8)
                : launch_fullk<1, FK_PLAIN, false>(fp, group_size, fp.g.NT, (hipStream_t)stream);
}

// residual_out = residual_in + fp16(xW + bias); ssq_out (optional): per-tile sums of residual_out^2, [M][ssq_ld >= N / 16]
extern "C" int mi355_gemm_fullk_residual(const void* gp, int wbits, int group_size, const void* residual_in, void* residual_out,
                                         float* ssq_out, int ssq_ld, mi355_stream_t stream) {
    FullKParams fp{};
    fp.g = *reinterpret_cast<const GemmParams*>(gp);
    FK_SET_STAMPS(fp);
    fp.ilv = TUNE(4) ? TUNE(4) - 1 : 1;
    if (!fullk_shape_ok(fp.g, wbits, group_size) || fp.g.N % 4 != 0) return MI355_ERR_UNSUPPORTED;
    fp.res_in = (const f16*)residual_in; fp.res_out = (f16*)residual_out; fp.ssq_out = ssq_out; fp.ssq_ld = ssq_ld;
    return launch_fullk<1, FK_RESID, false>(fp, group_size, fp.g.NT, (hipStream_t)stream);
}

// QKV projection (+ on-the-fly RMSNorm of its input) + bias + RoPE + Q extract + fp16 paged KV write (rows = tokens)
extern "C" int mi355_gemm_fullk_rope(const void* gp, int wbits, int group_size, const void* norm, const float* cos_sin, int32_t max_pos,
                                     const int32_t* positions, const int32_t* block_table, int32_t max_blocks_per_seq,
                                     int32_t q_len, int32_t nh, const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count,
                                     mi355_stream_t stream) {
    FullKParams fp{};
    fp.g = *reinterpret_cast<const GemmParams*>(gp);
    FK_SET_STAMPS(fp);
    fp.ilv = TUNE(4) ? TUNE(4) - 1 : 1;
    if (!fullk_shape_ok(fp.g, wbits, group_size)) return MI355_ERR_UNSUPPORTED;
    if (kv->kv_dtype != MI355_KV_FP16 || (kv->hd != 64 && kv->hd != 128)) return MI355_ERR_UNSUPPORTED;
    const int nheads = nh + 2 * kv->nkv;
    if (fp.g.N != nheads * kv->hd) return MI355_ERR_UNSUPPORTED;
    RopeEpi& r = fp.r;
    r.cos_sin = cos_sin; r.positions = positions; r.block_table = block_table; r.max_blocks = max_blocks_per_seq;
    r.nh = nh; r.nkv = kv->nkv; r.hd = kv->hd; r.page = kv->page; r.max_pos = max_pos; r.num_blocks = kv->num_blocks;
    r.q_len = q_len; r.oob_count = oob_count; r.kv_base = kv->kv_base; r.q_out = (f16*)q_out;
    if (norm) {
        set_norm(fp, (const mi355_fused_norm_t*)norm);
        return launch_fullk<2, FK_ROPE, true>(fp, group_size, nheads * (kv->hd / 32), (hipStream_t)stream);
    }
    return launch_fullk<2, FK_ROPE, false>(fp, group_size, nheads * (kv->hd / 32), (hipStream_t)stream);
}

// ---- the same two launches for 1-64 rows (the step driver: from 5) with the activations handed over as an image (gp->x: mi355_act_image_*), gemm_fullk64.hip
extern "C" int mi355_gemm_fullk_residual_img(const void* gp, int wbits, int group_size, const void* residual_in, void* residual_out,
                                             float* ssq_out, int ssq_ld, const void* norm_weight, float xg_scale, void* xg_img,
                                             mi355_stream_t stream) {
    FullKParams fp{};
    fp.g = *reinterpret_cast<const GemmParams*>(gp);
    if (!fullk64_shape_ok(fp.g, wbits, group_size) || fp.g.N % 4 != 0) return MI355_ERR_UNSUPPORTED;
    fp.res_in = (const f16*)residual_in; fp.res_out = (f16*)residual_out; fp.ssq_out = ssq_out; fp.ssq_ld = ssq_ld;
    fp.xg_img = (f16*)xg_img; fp.xg_gamma = (const f16*)norm_weight; fp.xg_scale = xg_scale;   // deferred RMSNorm of the produced rows (or null)
    fp.bf16 = fp.g.bf16;                                                                      // dtype of bias / residual / norm weight
    return mi355_gemm_fullk64(&fp, FK_RESID, group_size, stream);
}

// a row-parallel TP shard (O / down) straight into the rank's registered all-reduce buffer (FK_PUB): tgt = mi355_publish_target_t of internal.h
extern "C" int mi355_gemm_fullk_publish_img(const void* gp, int wbits, int group_size, const void* tgt_, mi355_stream_t stream) {
    const mi355_publish_target_t& tgt = *reinterpret_cast<const mi355_publish_target_t*>(tgt_);
    FullKParams fp{};
    fp.g = *reinterpret_cast<const GemmParams*>(gp);
    if (!fullk64_shape_ok(fp.g, wbits, group_size) || fp.g.N % 4 != 0) return MI355_ERR_UNSUPPORTED;
    fp.bf16 = fp.g.bf16;
    fp.pub_epoch = tgt.epoch; fp.pub_data = tgt.data; fp.pub_bytes = tgt.bytes; fp.pub_parity_elems = tgt.parity_elems; fp.pub_slot_elems = tgt.slot_elems;
    fp.pub_plain = tgt.plain_stores;
    return mi355_gemm_fullk64(&fp, FK_PUB, group_size, stream);
}

extern "C" int mi355_gemm_fullk_rope_img(const void* gp, int wbits, int group_size, const float* cos_sin, int32_t max_pos,
                                         const int32_t* positions, const int32_t* block_table, int32_t max_blocks_per_seq,
                                         int32_t q_len, int32_t nh, const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count,
                                         mi355_stream_t stream) {
    FullKParams fp{};
    fp.g = *reinterpret_cast<const GemmParams*>(gp);
    if (!fullk64_shape_ok(fp.g, wbits, group_size)) return MI355_ERR_UNSUPPORTED;
    // a 16-bit cache of the activation dtype (the epilogue stores the rotated K / V rows as they are; INT8 caches: rope_kv.hip)
    if (kv->kv_dtype != (fp.g.bf16 ? MI355_KV_BF16 : MI355_KV_FP16) || (kv->hd != 64 && kv->hd != 128)) return MI355_ERR_UNSUPPORTED;
    fp.bf16 = fp.g.bf16;
    const int nheads = nh + 2 * kv->nkv;
    if (fp.g.N != nheads * kv->hd) return MI355_ERR_UNSUPPORTED;
    RopeEpi& r = fp.r;
    r.cos_sin = cos_sin; r.positions = positions; r.block_table = block_table; r.max_blocks = max_blocks_per_seq;
    r.nh = nh; r.nkv = kv->nkv; r.hd = kv->hd; r.page = kv->page; r.max_pos = max_pos; r.num_blocks = kv->num_blocks;
    r.q_len = q_len; r.oob_count = oob_count; r.kv_base = kv->kv_base; r.q_out = (f16*)q_out;
    return mi355_gemm_fullk64(&fp, FK_ROPE, group_size, stream);
}
