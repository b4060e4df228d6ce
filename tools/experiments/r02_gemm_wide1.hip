// Weight-only W4 g128 GEMM for 32 < M <= 64 rows and wide N (gate_up), gfx950: ONE wave per SIMD, everything in registers.
//
// Same contract and weight image as gemm_wide.hip.  What bounds that kernel at 64 rows (profiles/r02_pmc_gemm_wide_m64.txt: a
// third of the wave cycles parked in s_waitcnt) is the depth of its weight ring: two waves per SIMD leave 256 registers
// each, 80 of them accumulators, 64 the activation fragments, and ONE chunk of weights (5 KB per wave, 40 KB per CU) in
// flight -- 16 units = 0.7 us of a ~2 us loaded HBM latency.  A second ring slot does not fit.
// Here a block is 4 waves, one per SIMD with 512 registers each, wave = K slice, and a wave owns ALL T = 10 tiles of the block:
//   * accumulators 160 AGPRs; weight ring TWO chunks deep (wr[k & 1][t] = tile t of chunk k, refilled with chunk k + 2
//     right after its last unit): 20 KB per wave, 80 KB per CU in flight;
//   * activation fragments of chunk k + 1 are loaded straight from L2 in MFMA B layout while chunk k is multiplied, into the
//     OTHER of two register sets -- one in VGPRs, one in AGPRs (an MFMA takes its B operand from either file), so there is no
//     copy, no LDS traffic and NO barrier anywhere in the main loop (a K slice has a single wave);
//   * every load is issued from inline asm and waited for with a hand-counted vmcnt (hipcc does not count asm loads, and its
//     own counts for loop-carried rings come out short), and every register that a load may have in flight sits in a FIXED
//     physical register (left to itself hipcc spills 76 registers of this kernel and inserts 467 v_accvgpr copies -- a copy of
//     a register whose load has not landed is silently wrong): constraint strings must be literals, so the operation stream is
//     generated (tools/gen_wide1.py -> gemm_wide1_phases.inc); operations per phase and the distances are spelled out below;
//   * the (4 MFMA + 13 VALU) unit is the fixed hand-ordered stream of gemm_wide.hip (WIDE_UNIT_W4); with one wave per SIMD
//     it runs at 52 instead of 43 ns (tools/probe/unit_rate.hip) -- the price of the registers.
#include "gemm_common.h"
#include <type_traits>

namespace {

struct Wide1Params {
    GemmParams g;
    int G; // tile groups (grid.x)
};

template <int T>
__global__ __launch_bounds__(256) void gemm_wide1_kernel(const Wide1Params wp) {
    const GemmParams& p = wp.g;
    constexpr int NKS = 4, MB = 4, NSUB = 1;                 // K slices = waves; 64 rows; one quantisation group per chunk (g128)
    constexpr int NU = 4 * T;                                // units per phase
    constexpr int XF = 4 * MB;                               // activation fragments per chunk, one load each (units 0 .. XF - 1)
    constexpr int LT = 1 + NSUB;                             // loads of one tile refill (unit 4 t + 3)
    constexpr int PT = XF + T * LT;                          // operations per phase
    // distances, in operations issued later, at the three kinds of wait (x load of unit u comes before that unit's refill):
    // (vmcnt is a 6-bit counter: a wave has at most 64 loads outstanding, and a distance above 63 is waited for as 63)
    constexpr int WX  = (T - (XF - 1) / 4) * LT;             // behind the last fragment load of a phase, at the next phase's start
    constexpr int WT0r = PT + (T - 1) * LT + (XF - 4);       // behind tile 0 of chunk k + 1 (unit 3 of phase k - 1) at unit NU - 1 of phase k
    constexpr int WT0 = WT0r > 63 ? 63 : WT0r;
    auto WT = [](int t) constexpr { const int n = PT + (T - 1) * LT + (XF - 4 * t - 4 > 0 ? XF - 4 * t - 4 : 0) + (XF < 4 * t ? XF : 4 * t); return n > 63 ? 63 : n; };
    constexpr uint32_t INV = 0x40000000u, INVX = 0x80000000u, FLAGS = 0x00020000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int t0 = (int)(((long)blockIdx.x * p.NT) / wp.G), t1 = (int)(((long)(blockIdx.x + 1) * p.NT) / wp.G);
    const int ntiles = t1 - t0;                              // <= T (host)
    const int c0 = blockIdx.y * p.cps, nch = min(p.cps, p.KC - c0);
    const int per = (nch + NKS - 1) / NKS;                   // phases of every wave of the block
    const int cw0 = c0 + ks * per;
    const int ncw = max(0, min(per, c0 + nch - cw0));

    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.qw, 0, p.qw_bytes, FLAGS);
    __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.meta, 0, p.meta_bytes, FLAGS);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, FLAGS);
    uint32_t toff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t ok = 0u - (uint32_t)(t < ntiles);
        toff[t] = (((uint32_t)(t0 + t) * (uint32_t)p.KC * 1024u) & ok) | (INV & ~ok);
    }
    const uint32_t lane16 = lane * 16u;
    const uint32_t mvoff = (uint32_t)(t0 * 16 + i) * 4u, mrow = (uint32_t)p.N_pad * 4u;
    const uint32_t xvoff = (uint32_t)((i * p.K + q * 8) * 2);   // + (row block 16 mb: 32 mb K bytes) + (k-step s: 64 s) + chunk: wave-uniform
    const uint32_t xrow = (uint32_t)p.K * 32u;                  // 16 rows down; rows >= M end up past x_bytes: zeros
    auto w_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * 1024u) & m) | (INV & ~m); };
    auto m_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * NSUB * mrow) & m) | (INV & ~m); };
    auto x_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * 256u) & m) | (INVX & ~m); };

    f32x4    acc[T][MB];
    u32x4    bv[MB][4], ba[MB][4];                           // B fragments [row block][k-step]: even chunks in VGPRs, odd chunks in AGPRs
    u32x4    wr[2][T];
    uint32_t mr[2][T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const W4Consts w4c = w4_consts();
    const f16x2 c960 = {(f16)960.f, (f16)960.f};
    f16x2 zn, znb, scl;
    auto meta_of = [&](uint32_t m) {
        zn  = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u));
        scl = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
        znb = zn + c960;
    };
    uint32_t wn, tmp;
    u32x4 aE, aO;
    // ---- prologue (generated: fixed register map, see tools/gen_wide1.py)
    {
        const uint32_t ws_a = w_soff(cw0, ncw > 0), ms_a = m_soff(cw0, ncw > 0);
        const uint32_t xs_b = x_soff(cw0, ncw > 0), ws_b = w_soff(cw0 + 1, ncw > 1), ms_b = m_soff(cw0 + 1, ncw > 1);
#define W1_PROLOGUE
#include "gemm_wide1_phases.inc"
#undef W1_PROLOGUE
    }
    meta_of(mr[0][0]);
    aE = __builtin_bit_cast(u32x4, dequant_w4_vc(wr[0][0][0], zn, znb, scl, w4c)); aO = aE;
    for (int k = 0; k < per; k += 2) {
        {
            const uint32_t xs = x_soff(cw0 + k + 1, k + 1 < ncw), ws = w_soff(cw0 + k + 2, k + 2 < ncw), ms = m_soff(cw0 + k + 2, k + 2 < ncw);
#define W1_PHASE0
#include "gemm_wide1_phases.inc"
#undef W1_PHASE0
        }
        if (k + 1 < per) {
            const uint32_t xs = x_soff(cw0 + k + 2, k + 2 < ncw), ws = w_soff(cw0 + k + 3, k + 3 < ncw), ms = m_soff(cw0 + k + 3, k + 3 < ncw);
#define W1_PHASE1
#include "gemm_wide1_phases.inc"
#undef W1_PHASE1
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");   // stragglers past the end; the last MFMAs' results are read by compiler code below

    // ---- merge the four K slices through LDS (TR tiles per round), epilogue
    constexpr int TR = 5;
    f32x4* red = reinterpret_cast<f32x4*>(smem);                  // [NKS][TR][MB][64]
#pragma unroll
    for (int r0 = 0; r0 < T; r0 += TR) {
        if (r0) __syncthreads();
#pragma unroll
        for (int t = r0; t < r0 + TR && t < T; ++t)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) red[((ks * TR + (t - r0)) * MB + mb) * 64 + lane] = acc[t][mb];
        __syncthreads();
        const int ntr = (T - r0 < TR) ? T - r0 : TR;
        for (int id = ks; id < ntr * MB; id += NKS) {
            const int tt = id / MB, mb = id - tt * MB, tb = r0 + tt;
            if (tb >= ntiles) continue;
            f32x4 v = red[((0 * TR + tt) * MB + mb) * 64 + lane];
#pragma unroll
            for (int s = 1; s < NKS; ++s) v += red[((s * TR + tt) * MB + mb) * 64 + lane];
            const int m = mb * 16 + i, n0 = (t0 + tb) * 16 + q * 4;
            if (m < p.M) gemm_store(p, v, m, n0, blockIdx.y);
        }
    }
}

} // namespace

// Direct mode only (fused epilogue in g.mode).  MI355_ERR_UNSUPPORTED: shape not taken, the caller falls back to gemm_wide.hip.
extern "C" int mi355_gemm_wide1(const void* gp, int wbits, int group_size, mi355_stream_t stream) {
    Wide1Params wp;
    wp.g = *reinterpret_cast<const GemmParams*>(gp);
    GemmParams& g = wp.g;
    constexpr int T = 10, CUS = 256;
    if (g.M <= 32 || g.M > 64 || wbits != 4 || group_size != 128 || g.mode == MODE_PARTIAL) return MI355_ERR_UNSUPPORTED;
    if (g.K % 128 != 0 || g.qw_bytes > 0x40000000u || g.meta_bytes > 0x40000000u || (uint64_t)g.M * g.K * 2 >= 0x40000000ull) return MI355_ERR_UNSUPPORTED;
    int G = (g.NT + T - 1) / T;
    if (G < CUS && g.NT >= CUS * (T - 3)) G = CUS;            // N alone fills the machine: spread the tiles over all CUs
    if (G < CUS * 3 / 4 || g.KC < 8) return MI355_ERR_UNSUPPORTED;
    g.cps = g.KC; g.nsplit = 1;
    wp.G = G;
    auto k = gemm_wide1_kernel<T>;
    constexpr size_t lds = (size_t)4 * 5 * 4 * 1024;          // merge buffer [NKS][TR][MB] KiB
    if (int e = raise_dynamic_lds((const void*)k, "gemm_wide1")) return e;
    hipLaunchKernelGGL(k, dim3(G, 1), dim3(256), lds, (hipStream_t)stream, wp);
    MI355_CHECK_LAUNCH("gemm_wide1_kernel");
    return MI355_OK;
}
