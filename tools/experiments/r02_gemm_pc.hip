// Weight-only W4 dequant GEMM for 32 < M <= 64 rows with SPECIALISED waves (producer / consumer), gfx950.
//
// Same contract and weight image as gemm.hip / gemm_wide.hip (reference slot: the W4A16 strategy of LinearFactory,
// rtp_llm/models_py/modules/factory/linear/factory.py:106-119).  Why another decomposition: at M = 64 the dequant costs
// 13 VALU per 8 weights against 4 MFMAs, and on gfx950 a v_mfma_f32_16x16x32_f16 blocks its OWN wave's issue for ~12 of
// its 16 cycles, so a wave that does both runs them back to back (gemm_wide: main loop 19.5 us for 8.3 us of matrix-pipe
// work and 7.3 us of VALU issue, profiles/r01_pmc_gemm_wide_m64.txt).  The two pipes of a SIMD do run concurrently when the
// instructions come from DIFFERENT waves.  So:
//   * a block = 12 waves = 4 K slices x (1 producer + 2 consumers); the three waves of a slice share a SIMD;
//   * the producer streams the slice's weights (<= 10 tiles per chunk, two chunks deep in registers, non-temporal 1 KiB
//     wave-loads), dequantises them (operand side, fp16) and parks the MFMA A-fragments in an LDS ring, one k-step of all
//     tiles (10 KiB) per batch, double-buffered -- a pure VALU / memory wave;
//   * each consumer owns half of the row blocks: it gathers its activation fragments of the NEXT chunk from global / L2 in
//     fragment layout while it works on the current ones (registers, ping-pong), reads every A-fragment of the batch once
//     from LDS and issues 2 MFMAs per fragment -- a matrix-pipe wave with 8 loads per chunk;
//   * one s_barrier per batch hands the ring over (producer one batch ahead); the K slices meet once at the end in LDS.
// Tail handling by buffer range checks only (offsets of absent tiles / chunks point past the buffer: loads return 0).
#include "gemm_common.h"

namespace {

struct PcParams {
    GemmParams g;
    int G; // tile groups (grid.x)
    int dbg;
    unsigned long long* stamps;   // tuning build: wall_clock64 per wave at start / loop entry / loop exit / end
};

template <int GS, int MB>
__global__ __launch_bounds__(768) void gemm_pc_kernel(const PcParams pp) {
    const GemmParams& p = pp.g;
    constexpr int NKS = 4, TB = 10;                       // K slices, max tiles per block
    constexpr int NSUB = 4 / GS, SPG = 4 / NSUB;          // GS = groups-per-chunk divisor: 4 -> g128, 2 -> g64, 1 -> g32
    constexpr int MBC = (MB + 1) / 2;                     // row blocks of consumer 0 (consumer 1 takes the rest)
    constexpr uint32_t INV = 0x40000000u, INVX = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* abuf = reinterpret_cast<u32x4*>(smem);                                  // [NKS][2][TB][64]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef MI355_TUNING
    unsigned long long st0 = wall_clock64(), st1 = 0, st2 = 0;
#endif
    const bool producer = wave < NKS;
    const int ks = producer ? wave : (wave - NKS) & 3;
    const int half = producer ? 0 : (wave - NKS) >> 2;
    const int i = lane & 15, q = lane >> 4;

    const int t0 = (int)(((long)blockIdx.x * p.NT) / pp.G), t1 = (int)(((long)(blockIdx.x + 1) * p.NT) / pp.G);
    const int ntiles = t1 - t0;                           // <= TB (host)
    const int c0  = blockIdx.y * p.cps;
    const int nch = min(p.cps, p.KC - c0);
    const int per = (nch + NKS - 1) / NKS;                // phases (chunks) of every slice of the block
    const int cw0 = c0 + ks * per;
    const int ncw = max(0, min(per, c0 + nch - cw0));

    u32x4* aring = abuf + (size_t)ks * 2 * TB * 64;       // this slice's ring: [2][TB][64]
    auto sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // raw: no vmcnt drain

    f32x4 acc[TB][2];                                     // consumers only
    constexpr uint32_t FLAGS = 0x00020000u;
    if (producer) {
        __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.qw, 0, p.qw_bytes, FLAGS);
        __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.meta, 0, p.meta_bytes, FLAGS);
        uint32_t toff[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const uint32_t ok = 0u - (uint32_t)(t < ntiles);
            toff[t] = (((uint32_t)(t0 + t) * (uint32_t)p.KC * 1024u) & ok) | (INV & ~ok);
#ifdef MI355_TUNING
            if (pp.dbg & 64) toff[t] = (((uint32_t)(t0 + t) * 1024u) & ok) | (INV & ~ok);          // chunk-major image (timing experiment only)
#endif
        }
        const uint32_t lane16 = lane * 16u;
        const uint32_t mvoff = (uint32_t)(t0 * 16 + i) * 4u, mrow = (uint32_t)p.N_pad * 4u;
#ifdef MI355_TUNING
        const uint32_t cstride = (pp.dbg & 64) ? (uint32_t)p.NT * 1024u : 1024u;
#else
        constexpr uint32_t cstride = 1024u;
#endif
        auto w_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * cstride) & m) | (INV & ~m); };
        auto m_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * NSUB * mrow) & m) | (INV & ~m); };

        // The producer's only VMEM traffic is this ring, so it is issued from inline asm and waited for with hand-counted
        // vmcnt: hipcc's own count at the loop header is a full drain (vmcnt(0) at the top of every second phase -- the refills
        // issued one batch earlier then cost an HBM round trip per phase; measured 43 us for gate_up at M = 64).  A tile's two
        // loads (weights, meta) are waited for exactly once, before its step-0 dequant: newer loads at that point are the
        // later tiles of the same ring (2 each) and the 2 * TB refills of the other ring.
        u32x4    wr[2][TB];                               // two chunks of weights in flight: ring[k & 1] holds chunk k
        uint32_t mr[2][TB];
        static_assert(NSUB == 1, "producer ring: one meta dword per (tile, chunk), i.e. group size 128");
        auto load_tile = [&](u32x4& w, uint32_t& m, int t, int c, bool valid) {
            const uint32_t ws = toff[t] + w_soff(c, valid), ms = m_soff(c, valid), mv = mvoff + t * 64u;
#ifdef MI355_TUNING
            if (pp.dbg & 32) {
                asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dword %1, %5, %6, %7 offen"
                             : "=&v"(w), "=&v"(m)
                             : "v"(lane16), "s"(rw), "s"(ws), "v"(mv), "s"(rm), "s"(ms)
                             : "memory");
                return;
            }
#endif
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen nt\n\tbuffer_load_dword %1, %5, %6, %7 offen"
                         : "=&v"(w), "=&v"(m)
                         : "v"(lane16), "s"(rw), "s"(ws), "v"(mv), "s"(rm), "s"(ms)
                         : "memory");
        };
        const W4Consts w4c = w4_consts();
        const f16x2 c960 = {(f16)960.f, (f16)960.f};
        auto dq = [&](const u32x4& w, uint32_t m, int s) -> u32x4 {
            const f16x2 zn = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u)), sc = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
            return __builtin_bit_cast(u32x4, dequant_w4_vc(w[s], zn, zn + c960, sc, w4c));
        };
        // ---- prologue: weights of chunks 0 and 1 -> rings, batch 0 -> ring slot 0
#pragma unroll
        for (int t = 0; t < TB; ++t) load_tile(wr[0][t], mr[0][t], t, cw0, ncw > 0);
#pragma unroll
        for (int t = 0; t < TB; ++t) load_tile(wr[1][t], mr[1][t], t, cw0 + 1, ncw > 1);
        static_for<0, TB>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(wr[0][t]), "+v"(mr[0][t]) : "i"(2 * (TB - 1 - t) + 2 * TB) : "memory");
            aring[(0 * TB + t) * 64 + lane] = dq(wr[0][t], mr[0][t], 0);
        });
        sync();                                           // prologue hand-over
#ifdef MI355_TUNING
        st1 = wall_clock64();
#endif
        // ---- steady state: in batch iteration b = 4 k + s the producer prepares batch b + 1 (step s + 1 of chunk k, or step
        // 0 of chunk k + 1) and refills a tile's registers with chunk k + 2 right after its last step was dequantised.
        for (int k = 0; k < per; k += 2) {
            static_for<0, 2>([&](auto rc) {
                constexpr int r = decltype(rc)::value;     // ring of chunk kk
                const int kk = k + r;
                if (kk < per) {
                    static_for<0, 4>([&](auto sc_) {
                        constexpr int s = decltype(sc_)::value;
                        constexpr int sn = (s + 1) & 3, rn = s == 3 ? (r ^ 1) : r;     // batch being prepared
                        const int par = (kk * 4 + s + 1) & 1;
                        static_for<0, TB>([&](auto tc) {
                            constexpr int t = decltype(tc)::value;
                            if constexpr (s == 3)      // first use of ring rn's tile t: its loads are TB - 1 - t tiles + one ring back
                                asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(wr[rn][t]), "+v"(mr[rn][t]) : "i"(2 * (TB - 1 - t) + 2 * TB) : "memory");
#ifdef MI355_TUNING
                            if (pp.dbg & 4) { if (!(pp.dbg & 8)) aring[(par * TB + t) * 64 + lane] = wr[rn][t]; }
                            else if (pp.dbg & 8) asm volatile("" :: "v"(dq(wr[rn][t], mr[rn][t], sn)));
                            else
#endif
                            aring[(par * TB + t) * 64 + lane] = dq(wr[rn][t], mr[rn][t], sn);
                            if constexpr (s == 2) load_tile(wr[r][t], mr[r][t], t, cw0 + kk + 2, kk + 2 < ncw && !(pp.dbg & 2));   // step 3 of ring r was just dequantised
                        });
                        sync();
                    });
                }
            });
        }
    } else {
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, FLAGS);
        auto x_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * 256u) & m) | (INVX & ~m); };
#pragma unroll
        for (int t = 0; t < TB; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
        const int mb0 = half * MBC;                       // first row block of this consumer
        const int nmb = half == 0 ? MBC : MB - MBC;       // 1 or 2 row blocks
        // fragment (m, s): rows 16 (mb0 + m) + i, k = 32 s + 8 q .. + 7 of the chunk; an absent second row block reads the first
        uint32_t xv[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) xv[m] = (uint32_t)((((mb0 + (m < nmb ? m : 0)) * 16 + i) * p.K + q * 8) * 2);
        u32x4 bq[2][2][4];                                // [ping-pong][row block][k-step]
        auto load_x = [&](int pp_, int c, bool valid) {
            const uint32_t so = x_soff(c, valid);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) bq[pp_][m][s] = bload128<0>(rx, xv[m] + s * 64u, so);
        };
        load_x(0, cw0, ncw > 0);
        sync();                                           // prologue hand-over
#ifdef MI355_TUNING
        st1 = wall_clock64();
#endif
        for (int k = 0; k < per; k += 2) {
            static_for<0, 2>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int kk = k + r;
                if (kk < per) {
                    if (!(pp.dbg & 1)) load_x(r ^ 1, cw0 + kk + 1, kk + 1 < ncw);       // next chunk's fragments while this one is multiplied
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int par = (kk * 4 + s) & 1;
                        // All TB fragments of the batch are read before the first MFMA needs one (PD reads ahead, then one
                        // read per tile's pair of MFMAs): left alone, hipcc funnels every tile through ONE fragment register
                        // and exposes the LDS latency ten times per batch (measured: 3700 cycles per batch, MFMA pipe 15 % busy).
#ifdef MI355_TUNING
                        if (pp.dbg & 16) { sync(); continue; }
#endif
                        u32x4 a[TB];
#pragma unroll
                        for (int t = 0; t < TB; ++t) a[t] = aring[(par * TB + t) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < TB; ++t) {
                            acc[t][0] = mfma16x16x32(__builtin_bit_cast(f16x8, a[t]), __builtin_bit_cast(f16x8, bq[r][0][s]), acc[t][0]);
                            acc[t][1] = mfma16x16x32(__builtin_bit_cast(f16x8, a[t]), __builtin_bit_cast(f16x8, bq[r][1][s]), acc[t][1]);   // absent row block: a copy of block 0, dropped at the store
                        }
                        constexpr int PD = 3;
                        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
                        for (int t = 0; t < TB - PD; ++t) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * PD, 0);
                        sync();
                    }
                }
            });
        }
    }
#ifdef MI355_TUNING
    st2 = wall_clock64();
    auto stamp_out = [&]() {
        if (lane == 0 && pp.stamps) { unsigned long long* d = pp.stamps + ((size_t)blockIdx.x * 12 + wave) * 4; d[0] = st0; d[1] = st1; d[2] = st2; d[3] = wall_clock64(); }
    };
#endif
    // ---- merge the K slices (slices 1..3 park their accumulators in LDS, slice 0's consumers add and store)
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);          // [3 slices][2 halves][TB][2][64]
    if (!producer && ks > 0) {
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m) red[((((ks - 1) * 2 + half) * TB + t) * 2 + m) * 64 + lane] = acc[t][m];
    }
    __syncthreads();
#ifdef MI355_TUNING
    if (producer || ks > 0) { stamp_out(); return; }
#else
    if (producer || ks > 0) return;
#endif
    const int mb0 = half * MBC, nmb = half == 0 ? MBC : MB - MBC;
#pragma unroll
    for (int t = 0; t < TB; ++t) {
        if (t >= ntiles) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m >= nmb) continue;
            f32x4 v = acc[t][m];
#pragma unroll
            for (int s = 0; s < 3; ++s) v += red[(((s * 2 + half) * TB + t) * 2 + m) * 64 + lane];
            const int row = (mb0 + m) * 16 + i, n0 = (t0 + t) * 16 + q * 4;
#ifdef MI355_TUNING
            if (pp.dbg & 128) { if (v[0] == 12345.f) gemm_store(p, v, row, n0, blockIdx.y); continue; }
#endif
            if (row < p.M) gemm_store(p, v, row, n0, blockIdx.y);
        }
    }
#ifdef MI355_TUNING
    stamp_out();
#endif
}

template <int GS, int MB>
int launch_pc_t(const PcParams& pp, hipStream_t st) {
    auto k = gemm_pc_kernel<GS, MB>;
    constexpr size_t ring_b = (size_t)4 * 2 * 10 * 1024, red_b = (size_t)3 * 2 * 10 * 2 * 1024;
    constexpr size_t lds = ring_b > red_b ? ring_b : red_b;
    if (int e = raise_dynamic_lds((const void*)k, "gemm_pc")) return e;
    hipLaunchKernelGGL(k, dim3(pp.G, pp.g.nsplit), dim3(768), lds, st, pp);
    MI355_CHECK_LAUNCH("gemm_pc_kernel");
    return MI355_OK;
}

} // namespace

// Plan + launch.  Returns the number of slabs written (partial mode), MI355_OK (direct mode), or MI355_ERR_UNSUPPORTED
// when the shape does not fit (the caller falls back to gemm_wide.hip / gemm.hip).
extern "C" int mi355_gemm_pc(const void* gp, int wbits, int group_size, int want_partial, int max_splits, mi355_stream_t stream) {
    GemmParams g = *reinterpret_cast<const GemmParams*>(gp);
    constexpr int TB = 10, CUS = 256;
    if (g.M <= 32 || g.M > 64 || wbits != 4) return MI355_ERR_UNSUPPORTED;
    if (group_size != 128) return MI355_ERR_UNSUPPORTED;   // g64 / g32 need 10-30 more meta registers than the 168 of a 3-wave SIMD: gemm_wide keeps them
    if (g.K % 128 != 0 || g.qw_bytes > 0x40000000u || g.meta_bytes > 0x40000000u) return MI355_ERR_UNSUPPORTED;
    PcParams pp;
    int G = (g.NT + TB - 1) / TB;
    int nsplit = 1;
    if (want_partial) {
        nsplit = CUS / G;
        if (nsplit > max_splits) nsplit = max_splits;
        if (nsplit > g.KC / 8) nsplit = g.KC / 8;       // >= two chunks per K slice
        if (nsplit < 1) nsplit = 1;
        if (G * nsplit < CUS * 3 / 4) return MI355_ERR_UNSUPPORTED;
    } else {
        if (G < CUS && g.NT >= CUS * (TB - 3)) G = CUS;
        if (G < CUS * 3 / 4) return MI355_ERR_UNSUPPORTED;
    }
    g.cps = (g.KC + nsplit - 1) / nsplit;
    g.nsplit = (g.KC + g.cps - 1) / g.cps;
    pp.g = g; pp.G = G; pp.dbg = TUNE(7);
#ifdef MI355_TUNING
    { extern unsigned long long* g_wide_stamps; pp.stamps = g_wide_stamps; }
#else
    pp.stamps = nullptr;
#endif
    hipStream_t st = (hipStream_t)stream;
    const int MB = (g.M + 15) / 16;
    int rc;
    rc = MB == 4 ? launch_pc_t<4, 4>(pp, st) : launch_pc_t<4, 3>(pp, st);
    if (rc != MI355_OK) return rc;
    return want_partial ? g.nsplit : MI355_OK;
}
