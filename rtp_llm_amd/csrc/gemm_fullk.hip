// Weight-only W4 (and fp16-weight) GEMM with the WHOLE reduction inside one block ("full K"), gfx950: no split-K slabs, no reduce launch,
// the consumer's elementwise work fused into the epilogue.
//
// The decode step is a chain of short dependent launches; each one pays a launch + ramp + drain of ~5 us around
// ~1-10 us of streaming (b = 1: 2.1 ms for 4.5 GB).  The split-K linears made that worse twice over: fp32 slabs through
// memory (qkv / o / down: 1.9-2.5x the weight bytes, profiles/r01_pmc_hbm_traffic.txt) and a second launch to fold them.
// This kernel gives one block ONE or TWO 16-column tiles and all of K: its waves are the K slices (wave w owns chunks
// [w KC / NW, (w+1) KC / NW)), activations go straight from L2 into MFMA B-fragments (every x element is used by exactly
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
#include "gemm_common.h"

namespace {

struct RopeEpi {
    const float*   cos_sin;
    const int32_t* positions;
    const int32_t* block_table;
    int            max_blocks, nh, nkv, hd, page, max_pos, num_blocks, q_len;
    int32_t*       oob_count;
    void*          kv_base;
    f16*           q_out;
};
struct FullKParams {
    GemmParams g;
    const f16* res_in;
    f16*       res_out;
    RopeEpi    r;
    // NORM: x is the un-normed residual row h; the kernel applies RMSNorm on the fly, x_n = gamma * fp16(h * rs), with
    // rs = rsqrt(sum_k h^2 / K + eps) rebuilt from the per-tile partial sums the producing launch left in ssq_in
    const float* ssq_in;     // [rows][ssq_ld], ssq_ld >= ssq_tiles: sum over the 16 columns of tile t of h[row]^2
    int          ssq_tiles, ssq_ld;
    const f16*   gamma;
    float        eps;
    float*       ssq_out;    // FK_RESID: the same partial sums of the rows this launch produces ([M][ssq_ld]), or null
};
enum { FK_PLAIN = 0, FK_RESID = 1, FK_ROPE = 2 };

template <int GS, int MB, int TPB, int EPI, bool NORM>
__global__ __launch_bounds__(MB >= 3 ? (TPB >= 2 ? 512 : 768) : 1024) void gemm_fullk_kernel(const FullKParams fp) {
    constexpr bool W16 = GS == 0;                    // GS: 4 -> W4 g128, 2 -> g64, 1 -> g32, 0 -> fp16 weights (no meta, 4 wave-loads per chunk)
    constexpr int LPC = W16 ? 4 : 1;
    constexpr int NSUB = W16 ? 1 : 4 / (W16 ? 1 : GS), SPG = 4 / NSUB;
    constexpr int WD = (GS == 0 && TPB >= 2) ? 1 : 2; // weight ring, chunks (fp16 tiles are 4x the registers)
    constexpr int HD = ((MB >= 3 && TPB < 2) || (MB == 2 && TPB >= 2) || NORM || TPB >= 4 || WD == 1) ? 2 : 4;   // activation ring, half chunks (2 k-steps x MB row blocks each); by register budget
    constexpr uint32_t FLAGS = 0x00020000u, OOBX = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);     // [NW][TPB * MB][64]
    const GemmParams& p = fp.g;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW   = blockDim.x >> 6;
    const int jj = lane & 15, q = lane >> 4;

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
    const int c0 = (wave * p.KC) / NW, c1 = ((wave + 1) * p.KC) / NW;
    const int n_ch = c1 - c0;

    __amdgpu_buffer_rsrc_t rw[TPB], rm[TPB];
#pragma unroll
    for (int t = 0; t < TPB; ++t) {
        const bool ok = tile[t] < p.NT;
        const char* wb = (const char*)p.qw + ((size_t)tile[t] * p.KC + c0) * (LPC * 1024);
        rw[t] = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, ok ? n_ch * LPC * 1024 : 0, FLAGS);
        const char* mb = (const char*)p.meta + ((size_t)c0 * NSUB * p.N_pad + tile[t] * 16) * 4;
        rm[t] = __builtin_amdgcn_make_buffer_rsrc((void*)(W16 ? (const char*)p.qw : mb), 0, (ok && n_ch > 0 && !W16) ? ((n_ch * NSUB - 1) * p.N_pad + 16) * 4 : 0, FLAGS);
    }
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, FLAGS);

    // B fragment (mb, chunk i, k-step s): row 16 mb + jj, k = 128 (c0 + i) + 32 s + 8 q .. + 7
    uint32_t xv[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) xv[mb] = (mb * 16 + jj < p.M) ? (uint32_t)(((size_t)(mb * 16 + jj) * p.K + c0 * 128 + q * 8) * 2) : OOBX;
    int nhc = 2 * n_ch;                              // half chunks of this wave; a VGPR so that the range select below is a
    asm volatile("" : "+v"(nhc));                    // v_cndmask, not a branch around the loads (which would drain vmcnt)

    float rs[MB];
    __amdgpu_buffer_rsrc_t rg = rx;
    if constexpr (NORM) rg = __builtin_amdgcn_make_buffer_rsrc((void*)fp.gamma, 0, p.K * 2, FLAGS);
    const uint32_t lane16 = lane * 16u, jj4 = jj * 4u;
    const uint32_t gv = (uint32_t)((c0 * 128 + q * 8) * 2);
    u32x4    gr[NORM ? HD : 1][2];                   // gamma of the half chunk's two k-steps (k = the fragment's 8 columns)
    u32x4    wr[WD][TPB][LPC];
    uint32_t mr[WD][TPB][NSUB];
    u32x4    xr[HD][MB][2];
    f32x4    acc[TPB][MB];
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_w = [&](int d, int i) {                // past the wave's range: out of the descriptor, returns 0, no traffic
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
#pragma unroll
            for (int lp = 0; lp < LPC; ++lp) wr[d][t][lp] = bload128<2 /*nt*/>(rw[t], lane16, (uint32_t)(i * LPC + lp) * 1024u);
            if constexpr (!W16) {
#pragma unroll
                for (int gi = 0; gi < NSUB; ++gi)
                    mr[d][t][gi] = __builtin_amdgcn_raw_buffer_load_b32(rm[t], jj4, (uint32_t)(i * NSUB + gi) * (uint32_t)p.N_pad * 4u, 0);
            }
        }
    };
    auto load_x = [&](int d, int hc) {               // past the range the next wave's slice would be read: force zeros
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) xr[d][mb][ss] = bload128<0>(rx, hc < nhc ? xv[mb] : OOBX, (uint32_t)hc * 128u + ss * 64u);
        if constexpr (NORM) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) gr[d][ss] = bload128<0>(rg, hc < nhc ? gv : OOBX, (uint32_t)hc * 128u + ss * 64u);
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
    auto half_chunk = [&](int d, int xb, int hf) {   // k-steps 2 hf, 2 hf + 1 of ring chunk d against activation buffer xb
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
                f16x8 a;
                if constexpr (W16) {
                    a = __builtin_bit_cast(f16x8, wr[d][t][s]);
                } else {
                    const uint32_t m = mr[d][t][s / SPG];
                    const f16x2 zn = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u)), sc = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
                    a = dequant_w4_vc(wr[d][t][0][s], zn, zn + c960, sc, w4c);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[t][mb] = mfma16x16x32(a, __builtin_bit_cast(f16x8, xr[xb][mb][ss]), acc[t][mb]);
            }
        }
    };

    // ---- the summing waves (wave mb < MB owns row block mb) request their epilogue operands now: residual / bias rows,
    // position -> block id -> rotation row would otherwise be two to three dependent round trips after the last barrier
    const int m_epi = wave * 16 + jj;                 // row of this lane in the epilogue (waves < MB only)
    const bool epi_wave = wave < MB && m_epi < p.M;
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
            const RopeEpi& R = fp.r;
            const int half = R.hd >> 1, hh = R.hd >> 5;
            const int h = tile[0] / (2 * hh), d0 = (tile[0] % (2 * hh)) * 16 + q * 4;
            if (p.bias) {
                pre_bias[0] = *reinterpret_cast<const f16x4*>(p.bias + h * R.hd + d0);
                pre_bias[TPB - 1] = *reinterpret_cast<const f16x4*>(p.bias + h * R.hd + d0 + half);
            }
            pre_pos = R.positions[m_epi];
            const int pos = min(max(pre_pos, 0), min(R.max_pos, R.max_blocks * R.page) - 1);
            pre_blk = R.block_table[(size_t)(m_epi / R.q_len) * R.max_blocks + pos / R.page];
            const float* cs = R.cos_sin + ((size_t)pos * half + d0) * 2;
            pre_cs[0] = *reinterpret_cast<const f32x4*>(cs); pre_cs[1] = *reinterpret_cast<const f32x4*>(cs + 4);
        }
    }
    // ---- everything the first round needs is requested before the first wait
#pragma unroll
    for (int d = 0; d < HD; ++d) load_x(d, d);
#pragma unroll
    for (int d = 0; d < WD; ++d) load_w(d, d);
    // NORM: 1 / rms of the rows from the producer's per-tile partial sums -- wave w adds up row w (one 16-byte load per lane,
    // one round trip under the weight loads just issued), the block meets once, every lane picks up its rows
    if constexpr (NORM) {
        float* rs_sh = reinterpret_cast<float*>(smem + (size_t)NW * TPB * MB * 1024);
        const int nv = fp.ssq_tiles >> 2;
        for (int r = wave; r < p.M; r += NW) {
            const f32x4* src = reinterpret_cast<const f32x4*>(fp.ssq_in + (size_t)r * fp.ssq_ld);
            float a = 0.f;
            for (int v = lane; v < nv; v += 64) { const f32x4 t = src[v]; a += (t[0] + t[1]) + (t[2] + t[3]); }
            a = wave_sum(a);
            if (lane == 0) rs_sh[r] = rsqrtf(a / (float)p.K + fp.eps);
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) rs[mb] = (mb * 16 + jj < p.M) ? rs_sh[mb * 16 + jj] : 0.f;
    }
    // rounds of WD chunks = 2 WD half chunks, no guards inside (slots past the end multiply zero activations)
    for (int i = 0; i < n_ch; i += WD) {
#pragma unroll
        for (int u = 0; u < 2 * WD; ++u) {
            const int hc = 2 * i + u;
            half_chunk(u >> 1, u % HD, u & 1);
            load_x(u % HD, hc + HD);
            if (u & 1) load_w(u >> 1, i + (u >> 1) + WD);
        }
    }

    // ---- the K slices meet in LDS; wave mb < MB sums row block mb of every tile of the block, in slice order
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) red[((size_t)wave * (TPB * MB) + t * MB + mb) * 64 + lane] = acc[t][mb];
    __syncthreads();
    if (wave >= MB) return;
    const int mb = wave;
    f32x4 v[TPB];
#pragma unroll
    for (int t = 0; t < TPB; ++t) {
        v[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < NW; ++w) v[t] += red[((size_t)w * (TPB * MB) + t * MB + mb) * 64 + lane];
    }
    const int m = mb * 16 + jj;
    if (m >= p.M) return;

    if constexpr (EPI == FK_PLAIN) {
#pragma unroll
        for (int t = 0; t < TPB; ++t)
            if (tile[t] < p.NT) gemm_store(p, v[t], m, tile[t] * 16 + q * 4, 0);
    } else if constexpr (EPI == FK_RESID) {
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

template <int GS, int MB, int TPB, int EPI, bool NORM>
int launch_fullk_t(const FullKParams& fp, int blocks, int NW, hipStream_t st) {
    auto k = gemm_fullk_kernel<GS, MB, TPB, EPI, NORM>;
    const size_t lds = (size_t)NW * TPB * MB * 1024 + (NORM ? 256 : 0);
    if (lds > 64 * 1024)
        if (int e = raise_dynamic_lds((const void*)k, "gemm_fullk")) return e;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), lds, st, fp);
    MI355_CHECK_LAUNCH("gemm_fullk_kernel");
    return MI355_OK;
}

// waves per block = K slices, bounded by the register budget of the shape and by `wave_cap` (callers that want several
// blocks per CU: 32 wave slots)
template <int TPB, int EPI, bool NORM>
int launch_fullk(const FullKParams& fp, int group_size, int blocks, hipStream_t st, int wave_cap = 16) {
    const GemmParams& g = fp.g;
    const int MB = g.M <= 16 ? 1 : (g.M <= 32 ? 2 : 4);
    if (NORM && MB > 1) return MI355_ERR_UNSUPPORTED;       // the on-the-fly norm is instantiated for <= 16 rows (latency regime)
    int maxw = MB >= 3 ? (TPB >= 2 ? 8 : 12) : 16;
    if (maxw > wave_cap) maxw = wave_cap;
    const int cpw = cdiv(g.KC, maxw), NW = cdiv(g.KC, cpw);
    if (NW < MB) return MI355_ERR_UNSUPPORTED;
#define FK_(GS_)                                                                                   \
    if constexpr (NORM || TPB >= 4) { if (MB > 1) return MI355_ERR_UNSUPPORTED; return launch_fullk_t<GS_, 1, TPB, EPI, NORM>(fp, blocks, NW, st); } \
    else switch (MB) {                                                                             \
    case 1: return launch_fullk_t<GS_, 1, TPB, EPI, false>(fp, blocks, NW, st);                    \
    case 2: return launch_fullk_t<GS_, 2, TPB, EPI, false>(fp, blocks, NW, st);                    \
    default: return launch_fullk_t<GS_, 4, TPB, EPI, false>(fp, blocks, NW, st);                   \
    }
    if (group_size == 128) { FK_(4) }
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

static void set_norm(FullKParams& fp, const mi355_fused_norm_t* n) {
    fp.ssq_in = n->tile_sumsq; fp.ssq_tiles = n->tiles; fp.ssq_ld = n->ld; fp.gamma = (const f16*)n->weight; fp.eps = n->eps;
}

// y = [RMSNorm](x) W (+ bias) with the epilogue of p.mode (fp16 / fp32 / SiLU-mul), one launch, no workspace.
// norm != null: x is the un-normed row, see FullKParams.  Wide outputs at a few rows (gate_up) take five tiles per block and
// two blocks per CU so that one wave of blocks covers the matrix.
extern "C" int mi355_gemm_fullk(const void* gp, int wbits, int group_size, const void* norm, mi355_stream_t stream) {
    FullKParams fp{};
    fp.g = *reinterpret_cast<const GemmParams*>(gp);
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
