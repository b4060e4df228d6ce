// Weight-only dequant GEMM for 16 < M <= 64 rows ("wide" decode batches), gfx950.
//
// Same contract and weight image as gemm.hip (reference slot: rtp_llm/models_py/modules/factory/linear/factory.py:106-119,
// W4A16 / W8A16 strategies) but a different decomposition.  At M = 64 the staged-x kernel of gemm.hip is bound by LDS
// B-fragment reads (every wave re-reads the whole 16 KiB x chunk for 1-2 KiB of weights), by its per-chunk barrier and
// by running on 148 of 256 CUs (BN = 256).  Here:
//
//  * one 8-wave block per CU (two waves per SIMD, <= 256 registers each) owns <= 2T adjacent 16-column tiles and a K
//    range.  Wave w = (K slice w & 3, tile half w >> 2): the four K slices split the block's chunks, the two halves
//    split its tiles; the two waves of a SIMD cover each other's dependency / memory stalls;
//  * a wave keeps the activations of its current chunk as MFMA B fragments in registers (MB x 4 fragments, 64 VGPRs at
//    MB = 4) and reuses them for its T tiles; the next chunk's fragments are gathered from global/L2 in fragment
//    layout by the two waves of the K slice (half each), parked in a double-buffered LDS region and read back right
//    after the last use of the old ones.  One s_barrier per chunk, no LDS traffic per tile;
//  * the fragment loads are issued a whole phase (T tiles) before they are written to LDS: vmcnt is an in-order
//    queue, so waiting for a young L2 load would also force every older HBM weight load to have landed (measured:
//    with a short wait distance the weight ring was effectively one tile deep);
//  * weights stream through a T-deep register ring (1 KiB wave-loads, non-temporal), dequantised on the operand side;
//    the (4 MFMA + 13 VALU) unit is a fixed hand-ordered instruction stream (WIDE_UNIT_W4): measured on MI355X a
//    16x16x32 MFMA blocks its own wave's issue for ~12 of its 16 cycles, so order and wave count decide the rate
//    (tools/probe/unit_rate.hip: 52 ns per unit with one wave per SIMD, 43 ns with two);
//  * the four K slices meet once in LDS at the end; split-K across blocks (fp32 slabs) only where N alone cannot fill
//    256 CUs (qkv / o / down).
//
// Tail handling is by buffer range checks only: offsets of tiles / chunks outside a wave's share are pushed past the
// end of the buffer (loads return 0 and cost nothing), so the main loop has no branches and every wave of a block
// runs the same number of phases (and barriers).
#include "gemm_common.h"
#include <type_traits>

namespace {

struct WideParams {
    GemmParams g;
    int G; // tile groups (grid.x)
    // deferred RMSNorm (mi355_deferred_norm_t): x holds gamma 2^-e h; the accumulators are multiplied by rsqrt(mean h^2 + eps) 2^e,
    // with sum h^2 of a row rebuilt from the producer's per-tile partial sums ssq[row * ssq_ld + t], t < ssq_tiles
    const float* ssq;
    int   ssq_tiles, ssq_ld;
    float eps, unscale;
    unsigned long long* stamps; // DBG & 4: wall_clock64 at start / after prologue / after the main loop / at the end, per wave
};

// LDS of the K-slice merge, ONE round at every row-block count: up to three row blocks every wave parks all its T x MB accumulator sets
// (<= 120 KB); at four (160 KB would not fit) a wave keeps the row block whose index is its K slice in registers -- it is the wave that sums
// that row block -- and parks the other three (120 KB)
constexpr size_t wide_merge_bytes(int MB, int T) { return (size_t)8 * T * (MB == 4 ? 3 : MB) * 1024; }

// T = tiles per wave; a block owns 2T tiles.  RING = chunks of weights a wave keeps requested ahead of the one it multiplies: 1 at 64 rows
// (a phase of 20 four-MFMA units outlasts the HBM latency; the registers are the B fragments'), 2 at <= 32 rows, where a phase is
// short, half the fragment registers are free and one chunk ahead (5 KB per wave, 40 KB per CU) left the loop waiting on memory.
template <int WBITS, int MB, int GS, int T, int DBG = 0, int RING = 1>
__global__ __launch_bounds__(512) void gemm_wide_kernel(const WideParams wp) {
    const GemmParams& p = wp.g;
    constexpr int NKS  = 4, NW = 8;                      // K slices, waves
    constexpr int LPC  = WBITS / 4;
    constexpr int NSUB = (GS > 0) ? 4 / GS : 1;
    constexpr int SPG  = 4 / NSUB;
    constexpr bool GROUPED = GS > 0;
    constexpr int NBL  = 4 * MB;                         // activation fragments per chunk
    constexpr int XF   = NBL / 2;                        // ... gathered by each of the two waves of a K slice
    constexpr int NU   = 4 * T;                          // (tile, k-step) units per chunk
    constexpr uint32_t INV  = 0x40000000u;               // images are <= 1 GiB: any sum with INV is out of range, no wrap
    constexpr uint32_t INVX = 0x80000000u;
    constexpr bool HAND = WBITS == 4;                   // hand-ordered unit (wide_unit_w4: WIDE_UNIT_W4 / _MB3 / _MB2 / _MB1)
    static_assert(XF <= NU - 8, "fragment writes, the barrier and the fragment reads must fit one phase");
    static_assert(RING == 1 || (RING == 2 && GS > 0), "two chunks ahead: group-wise instances only (per-channel meta lives in slot 0)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool KEEP = MB == 4 && !(DBG & 8);         // merge: see wide_merge_bytes (DBG & 8, tuning build: the two-round form of rounds 1-4, for A/B)
    constexpr int  TR = (MB == 4 && !KEEP) ? 3 : T;      // tiles per round
    constexpr size_t RS_OFF = (wide_merge_bytes(MB, T) > (size_t)2 * 4 * 4 * MB * 1024) ? wide_merge_bytes(MB, T) : (size_t)2 * 4 * 4 * MB * 1024;   // behind the stage / merge regions: 64 floats

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ks = wave & 3, th = wave >> 2;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if constexpr (DBG & 4) st0 = wall_clock64();
    const int i = lane & 15, q = lane >> 4;

    const int t0 = (int)(((long)blockIdx.x * p.NT) / wp.G), t1 = (int)(((long)(blockIdx.x + 1) * p.NT) / wp.G);
    const int ntiles = t1 - t0;                          // <= 2T (host)
    const int c0  = blockIdx.y * p.cps;
    const int nch = min(p.cps, p.KC - c0);
    const int per = (nch + NKS - 1) / NKS;               // phases of every wave of the block
    const int cw0 = c0 + ks * per;
    const int ncw = max(0, min(per, c0 + nch - cw0));    // chunks of this K slice

    constexpr uint32_t FLAGS = 0x00020000u;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.qw, 0, p.qw_bytes, FLAGS);
    __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.meta, 0, p.meta_bytes, FLAGS);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, FLAGS);

    uint32_t toff[T];                                    // wave-uniform byte offset of tile t's first chunk
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t ok = 0u - (uint32_t)(th * T + t < ntiles);
        toff[t] = (((uint32_t)(t0 + th * T + t) * (uint32_t)p.KC * (LPC * 1024u)) & ok) | (INV & ~ok);
    }
    const uint32_t lane16 = lane * 16u;
    const uint32_t mvoff  = (uint32_t)((t0 + th * T) * 16 + i) * 4u;  // + t * 64: meta of column 16 (t0 + th T + t) + i
    const uint32_t mrow   = (uint32_t)p.N_pad * 4u;
    uint32_t xvoff[XF];                                  // fragment j = th XF + jj: rows 16 (j / 4) + i, k-step j % 4
    const int MBLK = (p.M + 15) >> 4;                    // row blocks of an activation image
    const uint32_t xchunk = p.x_img ? (uint32_t)(4 * MBLK * 1024) : 256u;   // bytes from chunk to chunk
#pragma unroll
    for (int jj = 0; jj < XF; ++jj) {
        const int j = th * XF + jj;
        xvoff[jj] = (uint32_t)((((j / 4) * 16 + i) * p.K + q * 8) * 2 + (j % 4) * 64); // rows >= M: out of range
        // image: fragment (k-step, row block) is one dense 1 KB run (see common.h: gathered from the row-major tensor, 16 runs of 64 B)
        if (p.x_img) xvoff[jj] = (j / 4 < MBLK) ? (uint32_t)(((j % 4) * MBLK + j / 4) * 1024 + lane * 16) : INVX;
    }

    u32x4    wr[RING][T][LPC];                           // weight ring: slot k % RING holds tile t of chunk k, refilled with chunk k + RING
    uint32_t mr[RING][T][NSUB];
    f16x8    bq[MB][4];                                  // B fragments of the current chunk: [row block][k-step]
    u32x4    xt[XF];                                     // this wave's half of the fragments of chunk k + 2, in flight
    f32x4    acc[T][MB];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const W4Consts w4c = w4_consts();
    const f16x2 c960 = {(f16)960.f, (f16)960.f};

    // wave-uniform offsets of chunk `c` (or out of range when !valid); masks, not selects (see gemm_smallm.hip)
    auto w_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * (LPC * 1024u)) & m) | (INV & ~m); };
    auto m_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * NSUB * mrow) & m) | (INV & ~m); };
    auto x_soff = [&](int c, bool valid) { const uint32_t m = 0u - (uint32_t)valid; return (((uint32_t)c * xchunk) & m) | (INVX & ~m); };

    auto load_tile = [&](auto slot_c, int t, uint32_t ws, uint32_t ms) {
        constexpr int SL = decltype(slot_c)::value;
#pragma unroll
        for (int lp = 0; lp < LPC; ++lp) wr[SL][t][lp] = bload128<2 /*nt*/>(rw, lane16 + lp * 1024u, toff[t] + ws);
        if (GROUPED) {
#pragma unroll
            for (int gi = 0; gi < NSUB; ++gi)
                mr[SL][t][gi] = __builtin_amdgcn_raw_buffer_load_b32(rm, mvoff + t * 64u, ms + gi * mrow, 0);
        }
    };
    using Slot0 = std::integral_constant<int, 0>;
    using Slot1 = std::integral_constant<int, RING - 1>;

    // operand-side dequant of unit (t, s): column i of tile t, k = 32 s + 8 q .. + 7
    f16x2 zn, znb, scl;
    auto meta_of = [&](auto slot_c, int t, int s) {
        const uint32_t m = mr[GROUPED ? decltype(slot_c)::value : 0][t][GROUPED ? s / SPG : 0];
        zn  = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u));
        scl = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
        if (WBITS == 4) znb = zn + c960;
        if constexpr ((DBG & 512) != 0) {               // timing-only 9-VALU unit: the addends of its fma, -1024 s and -64 s (operand = s u: tame values, wrong results)
            const f16x2 km = {(f16)-1024.f, (f16)-1024.f}, kb = {(f16)-64.f, (f16)-64.f};
            zn = scl * km; znb = scl * kb;
        }
    };
    auto dq = [&](auto slot_c, int t, int s) -> f16x8 {
        constexpr int SL = decltype(slot_c)::value;
        if (s % SPG == 0) meta_of(slot_c, t, s);
        if (WBITS == 4) return dequant_w4_vc(wr[SL][t][0][s], zn, znb, scl, w4c);
        const u32x4 w = wr[SL][t][(s >> 1) % LPC];
        return dequant_w8<GROUPED>(w[(s & 1) * 2], w[(s & 1) * 2 + 1], zn, scl);
    };

    if (!GROUPED) { // per-channel: the meta row is constant along K
#pragma unroll
        for (int t = 0; t < T; ++t) mr[0][t][0] = __builtin_amdgcn_raw_buffer_load_b32(rm, mvoff + t * 64u, 0, 0);
    }

    // LDS: fragments of chunk c of K slice ks live in xbuf[c & 1][ks][fragment][lane]
    u32x4* xbuf = reinterpret_cast<u32x4*>(smem);
    auto xregion = [&](int par) { return xbuf + (par * NKS + ks) * (NBL * 64); };
    auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }; // raw: no vmcnt drain

    // ---- prologue: fragments of chunk 0 through LDS, chunk 1 in flight, weights of chunk 0 in the ring.  Issue order
    // as in the steady state (fragments of the later chunk first, then the ring): the compiler merges the vmcnt state of
    // this path into the loop head, and a shorter queue here would shorten every wait of the loop.
    {
        const uint32_t xs0 = x_soff(cw0, ncw > 0), xs1 = x_soff(cw0 + 1, !(DBG & 1) && ncw > 1);
        u32x4 x0t[XF];
#pragma unroll
        for (int jj = 0; jj < XF; ++jj) xt[jj] = bload128<0>(rx, xvoff[jj], xs1);
#pragma unroll
        for (int jj = 0; jj < XF; ++jj) x0t[jj] = bload128<0>(rx, xvoff[jj], xs0);
        // deferred norm: wave w sums the partial sums of rows 8 w .. 8 w + 7 (requested here, with the L2-resident fragments and in
        // front of the HBM weight requests; reduced below, once the fragments are parked)
        // (one batch of straight-line requests: a loop over rows or passes would wait for each load before asking for the next -- 8 round
        // trips, +3 us on the launch)
        f32x4 pq[8];
        const bool pq_on = wp.ssq && wave * 8 < p.M;                                // few rows: the waves past them ask for nothing
        if (pq_on) {
            __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)wp.ssq, 0, (uint32_t)((size_t)p.M * wp.ssq_ld * 4), FLAGS);
            const uint32_t lim = (uint32_t)wp.ssq_tiles * 4u;                       // bytes of a row's partial sums
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t rowoff = (uint32_t)((wave * 8 + r) * wp.ssq_ld) * 4u;   // rows >= M: past the descriptor
                pq[r] = __builtin_bit_cast(f32x4, bload128<0>(rq, lane * 16u < lim ? rowoff + lane * 16u : INVX));
            }
            if (lim > 1024u) {                                                       // K > 4096: tiles 256 .. 511 of every row
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint32_t rowoff = (uint32_t)((wave * 8 + r) * wp.ssq_ld) * 4u;
                    pq[r] += __builtin_bit_cast(f32x4, bload128<0>(rq, 1024u + lane * 16u < lim ? rowoff + 1024u + lane * 16u : INVX));
                }
            }
        }
        const uint32_t ws0 = w_soff(cw0, ncw > 0), ms0 = m_soff(cw0, ncw > 0);
#pragma unroll
        for (int t = 0; t < T; ++t) load_tile(Slot0{}, t, ws0, ms0);
        if constexpr (RING == 2) {
            const uint32_t ws1 = w_soff(cw0 + 1, !(DBG & 2) && ncw > 1), ms1 = m_soff(cw0 + 1, !(DBG & 2) && ncw > 1);
#pragma unroll
            for (int t = 0; t < T; ++t) load_tile(Slot1{}, t, ws1, ms1);
        }
        u32x4* x0 = xregion(0);
#pragma unroll
        for (int jj = 0; jj < XF; ++jj) x0[(th * XF + jj) * 64 + lane] = x0t[jj];
        block_sync();
#pragma unroll
        for (int j = 0; j < NBL; ++j) bq[j / 4][j % 4] = __builtin_bit_cast(f16x8, x0[j * 64 + lane]);
        if (pq_on) {
            float* rs_sh = reinterpret_cast<float*>(smem + RS_OFF);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float a = (pq[r][0] + pq[r][1]) + (pq[r][2] + pq[r][3]);
                a = dpp_add<0xB1>(a); a = dpp_add<0x4E>(a); a = dpp_add<0x141>(a); a = dpp_add<0x140>(a);   // 16-lane rows
                a = xor16_sum(a); a = xor32_sum(a);
                if (lane == 0) rs_sh[wave * 8 + r] = rsqrtf(a / (float)p.K + wp.eps) * wp.unscale;
            }
        }
    }
    f16x8 a_cur = dq(Slot0{}, 0, 0);
    // experiments (VERDICT r05 item 6 i): the second wave of every SIMD (th = 1: waves w and w + 4 share a SIMD) enters the loop HALF a unit (~44 cycles: 128) or a
    // whole unit (~90 cycles: 256, the control) behind its partner, so that one dequantises while the other's MFMAs run
    if constexpr (DBG & 128) { if (th == 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 11" ::: "memory"); }
    if constexpr (DBG & 256) { if (th == 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 9" ::: "memory"); }
    if constexpr (DBG & 64) { if (th == 1) __builtin_amdgcn_s_setprio(2); }   // experiment: ONE wave of every SIMD pair runs at a higher priority for the whole loop (asymmetric arbitration)
    if constexpr (DBG & 4) st1 = wall_clock64();
    u32x4 aE = __builtin_bit_cast(u32x4, a_cur), aO = aE;   // HAND: operand of even / odd units (fixed register tuples)

    // ---- phase k: chunk k of the K slice.  Units 0..XF-1 park the fragments of chunk k+1 (loaded one phase ago) in LDS
    // and fetch those of chunk k+2; the barrier follows; the last tile's units read chunk k+1 back into bq[..][s].
    auto phase = [&](auto slot_c, const int k) {
        constexpr int SL = decltype(slot_c)::value;                          // ring slot of chunk k
        using SlotN = std::integral_constant<int, (SL + 1) % RING>;          // ... of chunk k + 1 (the last unit dequantises its first dword)
        // DBG (timing experiments only): out-of-range offsets keep the instruction stream but remove the memory traffic
        const bool v1 = !(DBG & 2) && k + RING < ncw, v2 = !(DBG & 1) && k + 2 < ncw;
        const uint32_t ws = w_soff(cw0 + k + RING, v1), ms = m_soff(cw0 + k + RING, v1), xs = x_soff(cw0 + k + 2, v2);
        u32x4* xn = xregion((k + 1) & 1);
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int t = u / 4, s = u % 4;
            if constexpr (u < XF) {
                xn[(th * XF + u) * 64 + lane] = xt[u];
                xt[u] = bload128<0>(rx, xvoff[u], xs);
            }
            if constexpr (u == XF) block_sync();
            if constexpr (s == 3) load_tile(slot_c, t, ws, ms);             // dq(t, 3) was issued in unit (t, 2)
            constexpr int un = (u + 1) % NU, tn = un / 4, sn = un % 4;
            using SlotU = std::conditional_t<un == 0, SlotN, std::integral_constant<int, SL>>;
            if constexpr (HAND) {
                if constexpr (sn % SPG == 0) meta_of(SlotU{}, tn, sn);
                const uint32_t wn = wr[SlotU::value][tn][0][sn];
                wide_unit_w4<MB, u % 2 == 0, f16x8, (DBG & 32) ? 1 : (DBG & 512) ? 2 : 0>(aE, aO, wn, w4c, zn, znb, scl, acc[t][0], acc[t][MB > 1 ? 1 : 0], acc[t][MB > 2 ? 2 : 0],
                                             acc[t][MB > 3 ? 3 : 0], bq[0][s], bq[MB > 1 ? 1 : 0][s], bq[MB > 2 ? 2 : 0][s], bq[MB > 3 ? 3 : 0][s]);
            } else {
                const f16x8 a_next = dq(SlotU{}, tn, sn);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[t][mb] = mfma16x16x32(a_cur, bq[mb][s], acc[t][mb]);
                a_cur = a_next;
            }
            if constexpr (t == T - 1) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) bq[mb][s] = __builtin_bit_cast(f16x8, xn[(mb * 4 + s) * 64 + lane]);
            }
            // pin the memory operations of this unit where they were written (left free, the scheduler sinks a whole
            // phase of loads to its end and the ring reads then wait for loads issued a few cycles earlier)
            if constexpr (u < XF) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            if constexpr (s == 3) __builtin_amdgcn_sched_group_barrier(0x020, LPC + (GROUPED ? NSUB : 0), 0);
            if constexpr (!HAND) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, (13 + MB - 1) / MB + 1, 0); // its share of the dequant VALU
                }
            }
            if constexpr (t == T - 1) __builtin_amdgcn_sched_group_barrier(0x100, MB, 0);
            __builtin_amdgcn_sched_barrier(0);   // fence per unit: the groups above only order what is inside it
        });
    };
    for (int k = 0; k < per; k += RING) {
        phase(Slot0{}, k);
        if constexpr (RING == 2) if (k + 1 < per) phase(Slot1{}, k + 1);     // per is uniform over the block: its barriers stay matched
    }
    if constexpr (DBG & 4) st2 = wall_clock64();
    if constexpr (HAND) asm volatile("s_nop 15" ::: "memory"); // the last MFMAs' results are read by compiler code below
    __syncthreads(); // fragment regions are reused by the merge below

    // ---- merge the four K slices through LDS (one round), epilogue
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    // the summed set (tile tb of the block, row block mb): per-channel scale (W8), deferred RMSNorm, epilogue store
    auto finish = [&](f32x4 v, int tb, int mb) {
        const int m = mb * 16 + i, n0 = (t0 + tb) * 16 + q * 4;
        if (WBITS != 16 && !GROUPED) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t mm = __builtin_amdgcn_raw_buffer_load_b32(rm, (uint32_t)(n0 + r) * 4u, 0, 0);
                v[r] *= (float)as_h2(mm)[1];
            }
        }
        if (wp.ssq && m < p.M) v *= reinterpret_cast<const float*>(smem + RS_OFF)[m];   // deferred RMSNorm of row m (see WideParams)
        else if (p.x_img && p.bf16) v *= kImgBfUnscale;   // a plain image of a bf16 tensor holds x 2^-8 (common.h img_val; the deferred norm's has its own exponent)
        if (m < p.M) gemm_store(p, v, m, n0, blockIdx.y);
    };
    if constexpr (KEEP) {
        // wave (th, ks) parks the T x 3 sets of the row blocks != ks and sums row block ks of its half's T tiles: slices in order 0..3 as the
        // two-round form did (its own slice straight from the registers), so the bits are the same
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                if (mb != ks) red[((size_t)(wave * (T * 3) + t * 3 + (mb < ks ? mb : mb - 1))) * 64 + lane] = acc[t][mb];
        __syncthreads();
        auto sum_kept = [&](auto ks_c) {
            constexpr int KS = decltype(ks_c)::value;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int tb = th * T + t;
                if (tb >= ntiles) continue;
                f32x4 v;
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    const f32x4 term = (s == KS) ? acc[t][KS] : red[((size_t)((th * NKS + s) * (T * 3) + t * 3 + (KS < s ? KS : KS - 1))) * 64 + lane];
                    v = (s == 0) ? term : v + term;
                }
                finish(v, tb, KS);
            }
        };
        if (ks == 0) sum_kept(std::integral_constant<int, 0>{});
        else if (ks == 1) sum_kept(std::integral_constant<int, 1>{});
        else if (ks == 2) sum_kept(std::integral_constant<int, 2>{});
        else sum_kept(std::integral_constant<int, MB - 1>{});
    } else {
#pragma unroll
        for (int r0 = 0; r0 < T; r0 += TR) {
            if (r0) __syncthreads();
#pragma unroll
            for (int t = r0; t < r0 + TR && t < T; ++t)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) red[((wave * TR + (t - r0)) * MB + mb) * 64 + lane] = acc[t][mb];
            __syncthreads();
            const int ntr = (T - r0 < TR) ? T - r0 : TR;
            for (int id = wave; id < 2 * ntr * MB; id += NW) {
                const int h = id / (ntr * MB), rem = id - h * (ntr * MB), tt = rem / MB, mb = rem - tt * MB;
                const int tb = h * T + r0 + tt;          // tile inside the block
                if (tb >= ntiles) continue;
                f32x4 v = red[(((h * NKS + 0) * TR + tt) * MB + mb) * 64 + lane];
#pragma unroll
                for (int s = 1; s < NKS; ++s) v += red[(((h * NKS + s) * TR + tt) * MB + mb) * 64 + lane];
                finish(v, tb, mb);
            }
        }
    }
    if constexpr (DBG & 4) {
        if (lane == 0 && wp.stamps) {
            unsigned long long* d = wp.stamps + ((size_t)blockIdx.x * NW + wave) * 4;
            d[0] = st0; d[1] = st1; d[2] = st2; d[3] = wall_clock64();
        }
    }
}

template <int WBITS, int MB, int GS, int T, int DBG = 0, int RING = 1>
int launch_wide_t(const WideParams& wp, hipStream_t st) {
    auto k = gemm_wide_kernel<WBITS, MB, GS, T, DBG, RING>;
    constexpr size_t red_b = wide_merge_bytes(MB, T), stage_b = (size_t)2 * 4 * 4 * MB * 1024;
    constexpr size_t lds = (red_b > stage_b ? red_b : stage_b) + 256;   // + 1 / rms of the rows (deferred norm)
    if (int e = raise_dynamic_lds((const void*)k, "gemm_wide")) return e;
    hipLaunchKernelGGL(k, dim3(wp.G, wp.g.nsplit), dim3(512), lds, st, wp);
    MI355_CHECK_LAUNCH("gemm_wide_kernel");
    return MI355_OK;
}

} // namespace

#ifdef MI355_TUNING   // experiment switches of the tuning build only (tools/gemm_bench.py --var, tools/wide_stamps.py)
unsigned long long* g_wide_stamps = nullptr; // device buffer for DBG & 4 (mi355_debug_ptr)
int g_wide_dbg = 0; // 1 no activation reloads, 2 no weight refills, 3 both, 4 per-wave timestamps, +8 also for split-K shapes
extern "C" void mi355_debug_ptr(void* p) { g_wide_stamps = (unsigned long long*)p; }
#define WIDE_DBG g_wide_dbg
#define WIDE_STAMPS g_wide_stamps
#else
#define WIDE_DBG 0
#define WIDE_STAMPS nullptr
#endif

// Plan + launch.  Returns the number of slabs written (partial mode), MI355_OK (direct mode), or
// MI355_ERR_UNSUPPORTED when the shape does not fit this kernel (the caller falls back to gemm.hip).
static int gemm_wide_launch(const void* gp, int wbits, int group_size, int want_partial, int max_splits, const mi355_deferred_norm_t* dn,
                            mi355_stream_t stream);
// does the direct (one launch, no slabs) form take this linear at 1-64 rows?  N alone has to fill the chip (gate_up)
extern "C" int mi355_gemm_wide_direct_ok(const mi355_weight_t* w) {
    if (!w) return 0;                                // either activation dtype: the image entry (gemm.hip) decides
    if (!((w->wbits == 4 && (w->group_size == 128 || w->group_size == 64 || w->group_size == 32)) || (w->wbits == 8 && w->group_size == 0))) return 0;
    if (w->K % 128 != 0 || w->K_pad != w->K) return 0;
    const int NT = w->N_pad / 16, TB = 10, CUS = 256;
    int G = (NT + TB - 1) / TB;
    if (G < CUS && NT >= CUS * (TB - 3)) G = CUS;
    return G >= CUS * 3 / 4;
}
extern "C" int mi355_gemm_wide(const void* gp, int wbits, int group_size, int want_partial, int max_splits,
                               mi355_stream_t stream) {
    return gemm_wide_launch(gp, wbits, group_size, want_partial, max_splits, nullptr, stream);
}
// direct mode only, x an activation image (gp->x_img) and / or a deferred RMSNorm on the accumulators
extern "C" int mi355_gemm_wide_img(const void* gp, int wbits, int group_size, const mi355_deferred_norm_t* dn, mi355_stream_t stream) {
    return gemm_wide_launch(gp, wbits, group_size, 0, 1, dn, stream);
}
static int gemm_wide_launch(const void* gp, int wbits, int group_size, int want_partial, int max_splits, const mi355_deferred_norm_t* dn,
                            mi355_stream_t stream) {
    GemmParams g = *reinterpret_cast<const GemmParams*>(gp);
    constexpr int T = 5, TB = 2 * T, CUS = 256;     // tiles per wave / per block
    if (g.M < 1 || g.M > 64) return MI355_ERR_UNSUPPORTED;   // row-major callers come with > 16 rows (gemm.hip); fewer: the image entries only, on the two-row-block instances
    const bool w8 = wbits == 8 && group_size == 0;    // per-channel int8 (W8A16): compiler-scheduled unit, scale applied at the merge
    if (!w8 && !(wbits == 4 && (group_size == 128 || group_size == 64 || group_size == 32))) return MI355_ERR_UNSUPPORTED;
    if (g.K % 128 != 0 || g.qw_bytes > 0x40000000u || (!w8 && g.meta_bytes > 0x40000000u)) return MI355_ERR_UNSUPPORTED;
    WideParams wp;
    int G = (g.NT + TB - 1) / TB;                   // fewest groups with <= TB tiles each
    int nsplit = 1;
    // Split-K shapes: measured equal or behind the staged-x kernel at M = 64 (short K ranges leave 1-4 phases per wave
    // and the fixed prologue / merge dominates) and for short K at M <= 32 (qkv 9.0 vs 7.5 us); ahead for deep K at
    // M <= 32 (down 14.9 vs 20.0 us).  The rest stays on gemm.hip unless the experiment switch asks otherwise.
    if (want_partial && !(g.M <= 32 && g.KC >= 64) && WIDE_DBG < 8) return MI355_ERR_UNSUPPORTED;
    if (want_partial) {
        nsplit = CUS / G;
        if (nsplit > max_splits) nsplit = max_splits;
        if (nsplit > g.KC / 4) nsplit = g.KC / 4;   // >= one chunk per K slice
        if (nsplit < 1) nsplit = 1;
    } else {
        if (G < CUS && g.NT >= CUS * (TB - 3)) G = CUS;  // N alone fills the machine: spread tiles over all CUs
        if (G < CUS * 3 / 4) return MI355_ERR_UNSUPPORTED;
    }
    g.cps = (g.KC + nsplit - 1) / nsplit;
    g.nsplit = (g.KC + g.cps - 1) / g.cps;
    wp.g = g; wp.G = G; wp.stamps = WIDE_STAMPS;
    wp.ssq = nullptr; wp.ssq_tiles = wp.ssq_ld = 0; wp.eps = 0.f; wp.unscale = 1.f;
    if (dn) {
        if (want_partial) return MI355_ERR_UNSUPPORTED;   // the row factors are applied at the K-slice merge (after the per-channel scale of a W8 instance)
        wp.ssq = dn->tile_sumsq; wp.ssq_tiles = dn->tiles; wp.ssq_ld = dn->ld; wp.eps = dn->eps; wp.unscale = dn->unscale;
    }
    if (g.x_img && want_partial) return MI355_ERR_UNSUPPORTED;
    int rc;
    hipStream_t st = (hipStream_t)stream;
    const bool mb2 = g.M <= 32;
    const int mblk = (g.M + 15) >> 4;               // W4: an instance per row-block count (1 MFMA per unit at <= 16 rows ... 4 at 49-64); two
                                                    // chunks of weights ahead up to 32 rows (see the kernel)
#define WIDE_W4_(GS_) (mblk == 1 ? launch_wide_t<4, 1, GS_, T, 0, 2>(wp, st) : mblk == 2 ? launch_wide_t<4, 2, GS_, T, 0, 2>(wp, st) \
                       : mblk == 3 ? launch_wide_t<4, 3, GS_, T>(wp, st) : launch_wide_t<4, 4, GS_, T>(wp, st))
    if (w8)                    rc = mb2 ? launch_wide_t<8, 2, 0, T>(wp, st) : launch_wide_t<8, 4, 0, T>(wp, st);
    else if (group_size == 64) rc = WIDE_W4_(2);
    else if (group_size == 32) rc = WIDE_W4_(1);
#ifdef MI355_TUNING
    else if (mb2 && WIDE_DBG == 16) rc = launch_wide_t<4, 2, 4, T, 0, 1>(wp, st);   // one chunk ahead (the round-2..4 instance)
    else if (!mb2 && WIDE_DBG == 18) rc = launch_wide_t<4, 4, 4, T, 8>(wp, st);       // K-slice merge in two rounds (rounds 1-4)
    else if (mblk == 4 && WIDE_DBG == 32) rc = launch_wide_t<4, 4, 4, T, 32>(wp, st);   // round 6: s_setprio 2 over the unit's MFMA group
    else if (mblk == 4 && WIDE_DBG == 512) rc = launch_wide_t<4, 4, 4, T, 512>(wp, st);  // round 6, TIMING ONLY (results wrong): the 9-VALU unit of DESIGN 9 J
    else if (mblk == 4 && WIDE_DBG == 128) rc = launch_wide_t<4, 4, 4, T, 128>(wp, st);  // round 6: the second wave of every SIMD half a unit behind its partner
    else if (mblk == 4 && WIDE_DBG == 256) rc = launch_wide_t<4, 4, 4, T, 256>(wp, st);  // ... a whole unit behind (control)
    else if (mblk == 4 && WIDE_DBG == 64) rc = launch_wide_t<4, 4, 4, T, 64>(wp, st);   // round 6: the second wave of every SIMD at priority 2 for the whole loop
    else if (mb2 && WIDE_DBG == 17) rc = launch_wide_t<4, 2, 4, T, 0, 2>(wp, st);   // two row blocks whatever the row count
    else if (mb2 && WIDE_DBG == 2)  rc = launch_wide_t<4, 2, 4, T, 2, 2>(wp, st);   // instruction stream without weight traffic
    else if (mb2 && WIDE_DBG == 3)  rc = launch_wide_t<4, 2, 4, T, 3, 2>(wp, st);   // ... without any main-loop traffic
    else if ((WIDE_DBG & 7) != 0 && WIDE_DBG < 8)
        switch (WIDE_DBG & 7) {
            case 1: rc = launch_wide_t<4, 4, 4, T, 1>(wp, st); break;
            case 2: rc = launch_wide_t<4, 4, 4, T, 2>(wp, st); break;
            case 3: rc = launch_wide_t<4, 4, 4, T, 3>(wp, st); break;
            case 4: rc = launch_wide_t<4, 4, 4, T, 4>(wp, st); break;
            case 7: rc = launch_wide_t<4, 4, 4, T, 7>(wp, st); break;   // stamps of the traffic-free instruction stream
            default: rc = launch_wide_t<4, 4, 4, T>(wp, st);
        }
#endif
    else rc = WIDE_W4_(4);
#undef WIDE_W4_
    if (rc != MI355_OK) return rc;
    return want_partial ? g.nsplit : MI355_OK;
}
