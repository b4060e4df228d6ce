// EXPERIMENT, not built: hipcc (ROCm 7.2) allocates this kernel with 194 VGPRs + 256 AGPRs, 76 spills to scratch and 467
// v_accvgpr copies -- copies of ring registers whose asm-issued loads are still in flight would be silently wrong, so it was
// never run.  Kept as the written-down design (operation stream, vmcnt distances) for an all-assembly main loop.
//
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
//     own counts for loop-carried rings come out short): operations per phase and the distances are spelled out below;
//   * the (4 MFMA + 13 VALU) unit is the fixed hand-ordered stream of gemm_wide.hip (WIDE_UNIT_W4); with one wave per SIMD
//     it runs at 52 instead of 43 ns (tools/probe/unit_rate.hip) -- the price of the registers.
#include "gemm_common.h"
#include <type_traits>

namespace {

struct Wide1Params {
    GemmParams g;
    int G; // tile groups (grid.x)
};

// the unit with its B fragments in VGPRs (BC = "v") or AGPRs (BC = "a"); AIN/N0..N3 alternate between the two fixed tuples
#define W1_UNIT(EVEN, BC)                                                                                                   \
    do {                                                                                                                    \
        if constexpr (EVEN) {                                                                                               \
            asm volatile(WIDE_UNIT_W4("v[100:103]", "v104", "v105", "v106", "v107")                                         \
                         : [t] "=&v"(tmp), "=&{v[104:107]}"(aO), [c0] "+a"(acc[t][0]), [c1] "+a"(acc[t][1]),                \
                           [c2] "+a"(acc[t][2]), [c3] "+a"(acc[t][3])                                                       \
                         : "{v[100:103]}"(aE), [w] "v"(wn), [m0] "v"(w4c.m0), [m1] "v"(w4c.m1), [e0] "v"(w4c.e0),           \
                           [e1] "v"(w4c.e1), [zn] "v"(zn), [znb] "v"(znb), [sc] "v"(scl), [b0] BC(bc[0][s]),                \
                           [b1] BC(bc[1][s]), [b2] BC(bc[2][s]), [b3] BC(bc[3][s]));                                        \
        } else {                                                                                                            \
            asm volatile(WIDE_UNIT_W4("v[104:107]", "v100", "v101", "v102", "v103")                                         \
                         : [t] "=&v"(tmp), "=&{v[100:103]}"(aE), [c0] "+a"(acc[t][0]), [c1] "+a"(acc[t][1]),                \
                           [c2] "+a"(acc[t][2]), [c3] "+a"(acc[t][3])                                                       \
                         : "{v[104:107]}"(aO), [w] "v"(wn), [m0] "v"(w4c.m0), [m1] "v"(w4c.m1), [e0] "v"(w4c.e0),           \
                           [e1] "v"(w4c.e1), [zn] "v"(zn), [znb] "v"(znb), [sc] "v"(scl), [b0] BC(bc[0][s]),                \
                           [b1] BC(bc[1][s]), [b2] BC(bc[2][s]), [b3] BC(bc[3][s]));                                        \
        }                                                                                                                   \
    } while (0)

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
    auto ld_tile = [&](u32x4& w, uint32_t& m, int t, uint32_t ws, uint32_t ms) {
        const uint32_t so = toff[t] + ws, mv = mvoff + t * 64u;
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen nt\n\tbuffer_load_dword %1, %5, %6, %7 offen"
                     : "=&v"(w), "=&v"(m) : "v"(lane16), "s"(rw), "s"(so), "v"(mv), "s"(rm), "s"(ms) : "memory");
    };
    auto ld_xv = [&](u32x4& d, int j, uint32_t xs) {        // fragment j = 4 mb + s of a chunk
        const uint32_t so = xs + (uint32_t)(j / 4) * xrow + (uint32_t)(j % 4) * 64u;
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(d) : "v"(xvoff), "s"(rx), "s"(so) : "memory");
    };
    auto ld_xa = [&](u32x4& d, int j, uint32_t xs) {
        const uint32_t so = xs + (uint32_t)(j / 4) * xrow + (uint32_t)(j % 4) * 64u;
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&a"(d) : "v"(xvoff), "s"(rx), "s"(so) : "memory");
    };
    // one phase's operation stream without the arithmetic (the two pseudo-phases of the prologue): R = ring slot refilled,
    // fragments into the VGPR set (XA = false) or the AGPR set
    auto issue_phase = [&](auto rc, auto xac, uint32_t xs, uint32_t ws, uint32_t ms) {
        constexpr int R = decltype(rc)::value;
        constexpr bool XA = decltype(xac)::value;
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (u < XF) { if constexpr (XA) ld_xa(ba[u / 4][u % 4], u, xs); else ld_xv(bv[u / 4][u % 4], u, xs); }
            if constexpr (u % 4 == 3) ld_tile(wr[R][u / 4], mr[R][u / 4], u / 4, ws, ms);
        });
    };
    auto wait_tile = [&](u32x4& w, uint32_t& m, auto nc) {
        asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(w), "+v"(m) : "i"(decltype(nc)::value) : "memory");
    };

    // ---- prologue = phases -2 and -1 without arithmetic: (no fragments, tiles of chunk 0) then (fragments of chunk 0, tiles of chunk 1)
    issue_phase(std::integral_constant<int, 0>{}, std::true_type{}, INVX, w_soff(cw0, ncw > 0), m_soff(cw0, ncw > 0));
    issue_phase(std::integral_constant<int, 1>{}, std::false_type{}, x_soff(cw0, ncw > 0), w_soff(cw0 + 1, ncw > 1), m_soff(cw0 + 1, ncw > 1));
    wait_tile(wr[0][0], mr[0][0], std::integral_constant<int, WT0>{});
    meta_of(mr[0][0]);
    u32x4 aE = __builtin_bit_cast(u32x4, dequant_w4_vc(wr[0][0][0], zn, znb, scl, w4c)), aO = aE;

    // ---- phase k: chunk k from ring slot R = k & 1 and fragment set R; loads fragments of chunk k + 1 and tiles of chunk k + 2
    auto phase = [&](auto rc, int k) {
        constexpr int R = decltype(rc)::value;
        const uint32_t xs = x_soff(cw0 + k + 1, k + 1 < ncw), ws = w_soff(cw0 + k + 2, k + 2 < ncw), ms = m_soff(cw0 + k + 2, k + 2 < ncw);
        // this chunk's fragments were loaded during the previous phase
        if constexpr (R == 0) {
            asm volatile("s_waitcnt vmcnt(%c8)"
                         : "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[0][2]), "+v"(bv[0][3]), "+v"(bv[1][0]), "+v"(bv[1][1]), "+v"(bv[1][2]), "+v"(bv[1][3])
                         : "i"(WX) : "memory");
            asm volatile("" : "+v"(bv[2][0]), "+v"(bv[2][1]), "+v"(bv[2][2]), "+v"(bv[2][3]), "+v"(bv[3][0]), "+v"(bv[3][1]), "+v"(bv[3][2]), "+v"(bv[3][3]) :: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%c8)"
                         : "+a"(ba[0][0]), "+a"(ba[0][1]), "+a"(ba[0][2]), "+a"(ba[0][3]), "+a"(ba[1][0]), "+a"(ba[1][1]), "+a"(ba[1][2]), "+a"(ba[1][3])
                         : "i"(WX) : "memory");
            asm volatile("" : "+a"(ba[2][0]), "+a"(ba[2][1]), "+a"(ba[2][2]), "+a"(ba[2][3]), "+a"(ba[3][0]), "+a"(ba[3][1]), "+a"(ba[3][2]), "+a"(ba[3][3]) :: "memory");
        }
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int t = u / 4, s = u % 4;
            if constexpr (u < XF) { if constexpr (R == 0) ld_xa(ba[u / 4][u % 4], u, xs); else ld_xv(bv[u / 4][u % 4], u, xs); }
            if constexpr (s == 3) ld_tile(wr[R][t], mr[R][t], t, ws, ms);
            constexpr int un = (u + 1) % NU, tn = un / 4, sn = un % 4, rn = (u + 1 == NU) ? (R ^ 1) : R;
            if constexpr (sn == 0) {
                wait_tile(wr[rn][tn], mr[rn][tn], std::integral_constant<int, (u + 1 == NU) ? WT0 : WT(tn)>{});
                meta_of(mr[rn][tn]);
            }
            const uint32_t wn = wr[rn][tn][sn];
            uint32_t tmp;
            if constexpr (R == 0) { auto& bc = bv; W1_UNIT(u % 2 == 0, "v"); }
            else                  { auto& bc = ba; W1_UNIT(u % 2 == 0, "a"); }
            __builtin_amdgcn_sched_barrier(0);   // fence per unit: the order written IS the schedule
        });
    };
    for (int k = 0; k < per; k += 2) {
        phase(std::integral_constant<int, 0>{}, k);
        if (k + 1 < per) phase(std::integral_constant<int, 1>{}, k + 1);
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
