// Split-K weight-only W4 GEMM for 1-64 rows (the step driver: from 5) and DEEP K (down_proj: K = 18944, N = 3584), gfx950: fp32 slabs for the consumer's
// fold launch (mi355_add_rmsnorm), activations read as an image (mi355_act_image_*).
// Reference slot: LinearBase.forward of the W4A16 strategy (models_py/modules/factory/linear/linear_base.py:75-85,
// factory.py:106-119) for DenseMLP.down_proj (modules/hybrid/dense_mlp.py:95-106); the slabs are an internal hand-over.
//
// Why another kernel.  The staged kernel of gemm.hip runs this shape as 14 x 15 blocks: 15 fp32 slabs (13.8 MB written and read
// back for 36 MB of weights), a barrier and an LDS round trip per chunk, 19.6 us + a 6 us fold.  The bound of the shape at 64 rows is
// the (4 MFMA + 13 VALU) unit per (tile, k-step): 224 x 592 units = ~7 us on 224 CUs at the ~50 ns a SIMD needs per unit with two
// waves (tools/probe/unit_rate.hip, gemm_wide.hip), next to 6.5 us of HBM time.  So:
//   * a block owns FOUR adjacent 16-column tiles and a quarter of K (grid 56 x 4 = 224 blocks): 4 slabs instead of 15;
//   * its 8 waves (two per SIMD) are K slices of <= CPW chunks; a wave runs the units of ALL four tiles for its k-steps, so every
//     activation fragment (one dense 1 KB load from the image, common.h) feeds 16 MFMAs and is loaded by exactly one wave of the
//     block -- no LDS staging, no barrier in the main loop (vector-memory path per CU: 606 KB of activations + 151 KB of weights,
//     5 us at the ~150 GB/s a CU takes, under the issue time);
//   * weights stream through a two-chunk register ring per wave, activations through a ring of RING k-steps; the re-requests sit
//     right behind the last reader of their registers (vmcnt is in order: every wait then leaves the younger requests in flight);
//     Measured at 64 rows (profiles/r04_splitk64_variants.txt): 17.3 us; same instruction stream without activation traffic 15.0,
//     without weight traffic 12.5, without either 12.2 -- HBM-miss weight requests and L2-hit activation requests share the CU's
//     in-order memory pipeline, and together they cost more than the sum.  Deeper rings (3 chunks of weights + 4 k-steps of
//     activations, accumulators as VGPR operands so that hipcc allows 212 registers) measured the same (18.0 us);
//   * the unit is the fixed instruction stream of gemm_common.h (WIDE_UNIT_W4), operands carried in two fixed register tuples;
//   * the K slices meet once in LDS, each wave sums two (tile, row block) sets in slice order and stores them write-through (sc1)
//     into the slab of the block's K quarter.
#include "gemm_common.h"

namespace {

struct SplitK64Params {
    GemmParams g;     // x: activation image; partials: slabs [nsplit][M][N_pad]; cps: chunks per block
    int xmap;         // > 0: 1-D grid, XCDs per K split (see the kernel)
    int dbg;          // tuning build, timing only (same instruction stream): 1 no activation traffic, 2 no weight traffic
};

// Direct form (round 5, mi355_gemm_splitk64_direct): ONE K split and T = 2..5 tiles per block with the fused epilogue of the wide GEMM (bias / SiLU-mul,
// row-major or image output) -- for a column-parallel SHARD under tensor parallelism (gate_up of Qwen2-7B at tp 2: 1184 tiles, Llama-3-70B at tp 8: 448), whose
// N / tp columns leave the wide GEMM's 10-tiles-per-block form on half the chip (its phase structure needs >= 4 tiles per wave) and went through the staged
// split-K kernel + a fold launch: 19.0 + 5.3 us for 36 MB at tp 2.  Here every wave loads its own fragments, so few tiles per block cost nothing but the
// activation loads (T tiles share them).
// WB = 8 (round 5): per-channel INT8 weights -- two wave-loads per (tile, chunk), the operand is the exact integer u - 128 (4 v_perm +
// 4 v_pk_add per 8 weights: a lighter unit than W4's, left to the compiler's schedule), the column's scale multiplies the summed
// accumulators before the slab store (every K split is scaled alike, so the fold's sum of slabs is the scaled sum).
// DIRECT (round 6): the fused-epilogue store is an instance of its own -- with both stores behind a run-time `mode` the slab instances of the headline's
// down_proj carried gemm_store's code and registers and measured 17.2 -> 17.6 us (profiles/r06_r04_vs_r05_same_box.txt).
template <int WB, int GS, int MB, int T, int CPW, int RING, bool DIRECT = false, int UV = 0>   // UV: unit variant (2: the timing-only 9-VALU stream, tuning build)
__global__ __launch_bounds__(512) void gemm_splitk64_kernel(const SplitK64Params sp) {
    const GemmParams& p = sp.g;
    constexpr int NW = 8;
    constexpr bool W8 = WB == 8;
    constexpr int LPC = WB / 4;
    constexpr int NSUB = 4 / GS, SPG = 4 / NSUB;
    constexpr int NKS = CPW * 4;                         // k-steps of a wave (static schedule; past the slice: zero weights, zero activations)
    constexpr uint32_t FLAGS = 0x00020000u, OOBS = 0x40000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);         // [NW][T * MB][64]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jj = lane & 15, q = lane >> 4;
    // block -> (column group, K split).  Every block of a K split reads the same slice of the activation image: with the splits laid
    // out over the XCDs (block b runs on XCD b % 8 -- observed, used for speed only; sp.xmap = XCDs per split) an XCD's L2 fetches one
    // slice instead of the whole image (Qwen2-7B down at 64 rows: 4.8 MB instead of 19 MB crossing the fabric per launch)
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (sp.xmap > 0) { const int id = (int)blockIdx.x, xcd = id & 7, slot = id >> 3; by = xcd / sp.xmap; bx = slot * sp.xmap + xcd % sp.xmap; }
    const int t0 = bx * T;                               // first tile of the block
    const int cb = by * p.cps, ncb = min(p.cps, p.KC - cb);
    const int base = ncb / NW, rem = ncb - base * NW;    // waves 0 .. rem - 1 take one chunk more
    const int c0 = cb + wave * base + min(wave, rem);
    const int n_ch = base + (wave < rem ? 1 : 0);        // <= CPW (host)

    __amdgpu_buffer_rsrc_t rw[T], rm[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const bool ok = t0 + t < p.NT && n_ch > 0 && !(sp.dbg & 2);
        const char* wb = (const char*)p.qw + ((size_t)(t0 + t) * p.KC + c0) * (LPC * 1024);
        rw[t] = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, ok ? n_ch * LPC * 1024 : 0, FLAGS);
        const char* mb_ = (const char*)p.meta + (W8 ? (size_t)(t0 + t) * 16 * 4 : ((size_t)c0 * NSUB * p.N_pad + (t0 + t) * 16) * 4);
        rm[t] = __builtin_amdgcn_make_buffer_rsrc((void*)mb_, 0, (t0 + t < p.NT) ? (W8 ? 64 : (ok ? ((n_ch * NSUB - 1) * p.N_pad + 16) * 4 : 0)) : 0, FLAGS);
    }
    const int MBLK = (p.M + 15) >> 4;                    // row blocks of the image (<= MB)
    const char* xb = (const char*)p.x + (size_t)c0 * 4 * MBLK * 1024;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (sp.dbg & 1) ? 0u : (uint32_t)(n_ch * 4 * MBLK * 1024), FLAGS);
    const uint32_t lane16 = lane * 16u, jj4 = jj * 4u;

    u32x4    wr[2][T][LPC];                              // weight ring: chunk c in slot c & 1
    uint32_t mr[2][T][NSUB];
    u32x4    xr[RING][MB];                               // activation ring: k-step ks in slot ks % RING
    auto load_w = [&](int c) {                           // c compile-time at every call site
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int lp = 0; lp < LPC; ++lp) wr[c & 1][t][lp] = bload128<2 /*nt*/>(rw[t], lane16, (uint32_t)(c * LPC + lp) * 1024u);
            if constexpr (!W8) {
#pragma unroll
                for (int gi = 0; gi < NSUB; ++gi)
                    mr[c & 1][t][gi] = __builtin_amdgcn_raw_buffer_load_b32(rm[t], jj4, (uint32_t)(c * NSUB + gi) * (uint32_t)p.N_pad * 4u, 0);
            }
        }
    };
    u32x4 sc4[T];                                        // W8: (zero, scale) words of this lane's four columns per tile, for the slab store
    if constexpr (W8) {
#pragma unroll
        for (int t = 0; t < T; ++t) sc4[t] = bload128<0>(rm[t], (uint32_t)q * 16u);
    }
    auto load_x = [&](int ks) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            xr[ks % RING][mb] = bload128<0>(rx, lane16, mb < MBLK ? (uint32_t)((ks * MBLK + mb) * 1024) : OOBS);
    };
    // request order = need order (fenced: left free, hipcc sinks the 4-byte (zero, scale) loads behind the 16-byte ones and the first
    // unit's wait for them drains the whole prologue)
    load_w(0);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, RING>([&](auto k_) { load_x(decltype(k_)::value); });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (CPW > 1) load_w(1);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[T][MB];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const W4Consts w4c = w4_consts();
    const f16x2 c960 = {(f16)960.f, (f16)960.f};
    const f16x2 zn8 = {(f16)-1152.f, (f16)-1152.f};     // W8: byte u under the exponent of 1024, minus 1024 + 128
    // (zero, scale) of the group in use PER TILE: consecutive units belong to different tiles (k-step-major order), so a tile's words
    // live in its own registers and are refreshed when its next unit starts a new group
    f16x2 zn[T], znb[T], scl[T];
    auto meta_of = [&](int c, int t, int s) {
        const uint32_t m = mr[c & 1][t][s / SPG];
        zn[t]  = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u));
        scl[t] = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
        znb[t] = zn[t] + c960;
        if constexpr (UV == 2) {                         // timing-only 9-VALU unit: the addends of its fma, -1024 s and -64 s (operand = s u: tame values, wrong results)
            const f16x2 km = {(f16)-1024.f, (f16)-1024.f}, kb = {(f16)-64.f, (f16)-64.f};
            zn[t] = scl[t] * km; znb[t] = scl[t] * kb;
        }
    };
    u32x4 aE, aO;                                        // W4: operand of even / odd units (fixed register tuples)
    if constexpr (!W8) {
        meta_of(0, 0, 0);
        aE = __builtin_bit_cast(u32x4, dequant_w4_vc(wr[0][0][0][0], zn[0], znb[0], scl[0], w4c)); aO = aE;
    }

    // ---- units in k-step-major order: u = (ks * T + t), ks = 4 c + s
    constexpr int NU = NKS * T;
    auto unit = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int ks = u / T, t = u % T, c = ks / 4, s = ks % 4;
        if constexpr (W8) {
            const u32x4 w = wr[c & 1][t][s >> 1];        // wave-load s / 2 of the chunk: k-steps 2 (s / 2), + 1, two dwords each
            const f16x8 a = dequant_w8<false>(w[(s & 1) * 2], w[(s & 1) * 2 + 1], zn8, zn8);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[t][mb] = mfma16x16x32(a, __builtin_bit_cast(f16x8, xr[ks % RING][mb]), acc[t][mb]);
        } else {
        constexpr int un = u + 1, ksn = un / T, tn = un % T, cn = ksn / 4, sn = ksn % 4;   // the unit whose operand this one prepares
        uint32_t wn = 0;
        if constexpr (un < NU) {
            if constexpr (sn % SPG == 0) meta_of(cn, tn, sn);
            wn = wr[cn & 1][tn][0][sn];
        }
        wide_unit_w4<MB, u % 2 == 0, u32x4, UV>(aE, aO, wn, w4c, zn[tn % T], znb[tn % T], scl[tn % T], acc[t][0], acc[t][MB > 1 ? 1 : 0], acc[t][MB > 2 ? 2 : 0],
                                     acc[t][MB > 3 ? 3 : 0], xr[ks % RING][0], xr[ks % RING][MB > 1 ? 1 : 0], xr[ks % RING][MB > 2 ? 2 : 0],
                                     xr[ks % RING][MB > 3 ? 3 : 0]);
        }
        // re-requests right behind the last reader of their registers: the k-step's fragments after its last tile; the chunk's
        // weights after the unit that prepared the last operand taken from them
        if constexpr (t == T - 1 && ks + RING < NKS) load_x(ks + RING);
        if constexpr (s == 3 && t == T - 1 && c + 2 < CPW) load_w(c + 2);
        __builtin_amdgcn_sched_barrier(0);               // fence per unit: keeps the requests where they were written
    };
    // the slices of a block differ by at most one chunk (base / base + 1): the waves without the last chunk skip its units (zero
    // weights otherwise: 48 of a block's 640 units at K = 18944)
    constexpr int NU_HEAD = (CPW - 1) * 4 * T;
    static_for<0, NU_HEAD>(unit);
    if (n_ch == CPW) static_for<NU_HEAD, NU>(unit);
    if constexpr (!W8) asm volatile("s_nop 15" ::: "memory");   // the last MFMAs' results are read by compiler code below

    // ---- the K slices meet in LDS; wave w sums the sets e = w, w + 8, ... (set e = tile e / MB, row block e % MB) and stores them
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) red[((size_t)wave * (T * MB) + t * MB + mb) * 64 + lane] = acc[t][mb];
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = slab_rsrc(p, p.nsplit);
    for (int e = wave; e < T * MB; e += NW) {
        const int t = e / MB, mb = e % MB, m = mb * 16 + jj;
        f32x4 v = red[((size_t)e) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[((size_t)w * (T * MB) + e) * 64 + lane];
        if constexpr (W8) {
            static_for<0, T>([&](auto t_) {              // this set's tile: its scales live in registers indexed at compile time
                if (decltype(t_)::value == t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= (float)as_h2(sc4[decltype(t_)::value][r])[1];
                }
            });
        }
        if (p.bf16) v *= kImgBfUnscale;                  // the image of a bf16 tensor holds x 2^-8 (common.h img_val)
        if (m < p.M && t0 + t < p.NT) {
            if constexpr (!DIRECT) st_slab(rs, (uint32_t)((((size_t)by * p.M + m) * p.N_pad + (t0 + t) * 16 + q * 4) * 4), v);
            else gemm_store(p, v, m, (t0 + t) * 16 + q * 4, 0);   // direct form (one K split): bias / fp16 / SiLU-mul store, row-major or image
        }
    }
}

template <int WB, int GS, int MB, int T, int CPW, int RING, bool DIRECT = false, int UV = 0>
int launch_splitk64_t(const SplitK64Params& sp, int G, hipStream_t st) {
    auto k = gemm_splitk64_kernel<WB, GS, MB, T, CPW, RING, DIRECT, UV>;
    const size_t lds = (size_t)8 * T * MB * 1024;
    if (lds > 160 * 1024) return MI355_ERR_UNSUPPORTED;
    if (lds > 64 * 1024)
        if (int e = raise_dynamic_lds((const void*)k, "gemm_splitk64")) return e;
    if (sp.xmap > 0) hipLaunchKernelGGL(k, dim3(G * sp.g.nsplit), dim3(512), lds, st, sp);
    else             hipLaunchKernelGGL(k, dim3(G, sp.g.nsplit), dim3(512), lds, st, sp);
    MI355_CHECK_LAUNCH("gemm_splitk64_kernel");
    return MI355_OK;
}

} // namespace

// Plan: four tiles per block; the fewest K splits that put a block on >= 3/4 of the CUs, each wave <= 5 chunks.
// Returns the number of slabs, or MI355_ERR_UNSUPPORTED (shape not deep / wide enough: the caller stays on gemm.hip).
extern "C" int mi355_gemm_splitk64_plan(int M, int NT, int KC, int wbits, int group_size, int max_splits, int* cps_out) {
    if (M < 1 || M > 64 || !((wbits == 4 && (group_size == 128 || group_size == 64 || group_size == 32)) || (wbits == 8 && group_size == 0))) return MI355_ERR_UNSUPPORTED;
    const int G = (NT + 3) / 4;
    int ns = (192 + G - 1) / G;                          // >= 192 blocks
    if (ns < 2) return MI355_ERR_UNSUPPORTED;            // N alone fills the chip: the wide kernel's shape
    if (ns > max_splits) ns = max_splits;
    int cps = (KC + ns - 1) / ns;
    while (ns < max_splits && (cps + 7) / 8 > 5) { ++ns; cps = (KC + ns - 1) / ns; }
    ns = (KC + cps - 1) / cps;
    if ((cps + 7) / 8 > 5 || cps < 8 || G * ns > 512) return MI355_ERR_UNSUPPORTED;   // every wave >= 1 chunk, <= 5
    if (cps_out) *cps_out = cps;
    return ns;
}

// Direct form: tiles per block T in 5..2 such that the blocks fill 5/8 .. all of the 256 CUs in ONE round, every wave <= 5 chunks (T >= 3) or <= 8 (T = 2).
// Returns T, or MI355_ERR_UNSUPPORTED (W4 g128, 1-64 rows only).
extern "C" int mi355_gemm_splitk64_direct_plan(int M, int NT, int KC, int wbits, int group_size) {
    if (M < 1 || M > 64 || wbits != 4 || group_size != 128 || KC < 8) return MI355_ERR_UNSUPPORTED;
    const int cpw = (KC + 7) / 8;
    for (int T = 5; T >= 2; --T) {
        const int G = (NT + T - 1) / T;
        if (G >= 160 && G <= 256 && cpw <= (T == 2 ? 8 : 5)) return T;
    }
    return MI355_ERR_UNSUPPORTED;
}

// gp: GemmParams with x = activation image, mode / y / bias / ldy / y_img of a direct linear (not MODE_PARTIAL)
extern "C" int mi355_gemm_splitk64_direct(const void* gp, int wbits, int group_size, mi355_stream_t stream) {
    SplitK64Params sp;
    sp.g = *reinterpret_cast<const GemmParams*>(gp);
    sp.dbg = 0; sp.xmap = 0;
    GemmParams& g = sp.g;
    if (g.K != g.KC * 128 || g.mode == MODE_PARTIAL || !g.y) return MI355_ERR_UNSUPPORTED;
    const int T = mi355_gemm_splitk64_direct_plan(g.M, g.NT, g.KC, wbits, group_size);
    if (T < 0) return T;
    g.nsplit = 1; g.cps = g.KC;
    const int G = (g.NT + T - 1) / T, cpw = (g.KC + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    const int mblk = (g.M + 15) >> 4;
    // activation ring: 3 k-steps, 2 where T x MB accumulators + a two-chunk weight ring of T tiles leave no room for the third
#define SKD_T_(T_, CPW_, RING_) (mblk == 1 ? launch_splitk64_t<4, 4, 1, T_, CPW_, RING_, true>(sp, G, st) : mblk == 2 ? launch_splitk64_t<4, 4, 2, T_, CPW_, RING_, true>(sp, G, st) \
                                 : mblk == 3 ? launch_splitk64_t<4, 4, 3, T_, CPW_, RING_, true>(sp, G, st) : launch_splitk64_t<4, 4, 4, T_, CPW_, RING_, true>(sp, G, st))
    // CPW = the chunks of the longest slice exactly (waves one chunk short skip the last chunk's units; a larger CPW would run them on zeros)
    if (T == 5) return cpw <= 4 ? SKD_T_(5, 4, 2) : SKD_T_(5, 5, 2);
    if (T == 4) return cpw <= 3 ? SKD_T_(4, 3, 3) : cpw == 4 ? SKD_T_(4, 4, 3) : SKD_T_(4, 5, 3);
    if (T == 3) return cpw <= 4 ? SKD_T_(3, 4, 3) : SKD_T_(3, 5, 3);
    return cpw <= 5 ? SKD_T_(2, 5, 3) : SKD_T_(2, 8, 2);   // eight-chunk slices: two k-steps of activations in flight (three spill at four row blocks)
#undef SKD_T_
}

// gp: GemmParams with x = activation image, partials = slabs; returns the number of slabs written.
extern "C" int mi355_gemm_splitk64(const void* gp, int wbits, int group_size, int max_splits, mi355_stream_t stream) {
    SplitK64Params sp;
    sp.g = *reinterpret_cast<const GemmParams*>(gp);
    sp.dbg = TUNE(7);
    GemmParams& g = sp.g;
    if (g.K != g.KC * 128 || !g.partials) return MI355_ERR_UNSUPPORTED;
    int cps = 0;
    const int ns = mi355_gemm_splitk64_plan(g.M, g.NT, g.KC, wbits, group_size, max_splits, &cps);
    if (ns < 0) return ns;
    g.nsplit = ns; g.cps = cps; g.mode = MODE_PARTIAL;
    const int G = (g.NT + 3) / 4, cpw = (cps + 7) / 8;
    sp.xmap = (8 % ns == 0 && G % (8 / ns) == 0 && (G * ns) % 8 == 0) ? 8 / ns : 0;   // K splits over whole XCDs when the counts divide
    hipStream_t st = (hipStream_t)stream;
    const int mblk = (g.M + 15) >> 4;                    // row blocks: an instance per count (1 MFMA per unit at <= 16 rows ... 4 at 49-64)
    int rc;
#define SK_MB_(WB_, GS_, MB_) (cpw <= 3 ? launch_splitk64_t<WB_, GS_, MB_, 4, 3, 3>(sp, G, st) : launch_splitk64_t<WB_, GS_, MB_, 4, 5, 3>(sp, G, st))
#define SK_(WB_, GS_) rc = mblk == 1 ? SK_MB_(WB_, GS_, 1) : mblk == 2 ? SK_MB_(WB_, GS_, 2) : mblk == 3 ? SK_MB_(WB_, GS_, 3) : SK_MB_(WB_, GS_, 4)
#ifdef MI355_TUNING
    if (wbits == 4 && group_size == 128 && mblk == 4 && cpw > 3 && sp.dbg == 16) {   // round 6, TIMING ONLY (results wrong): the 9-VALU unit of DESIGN 9 J (--debug-set 7=16; NOT 9: bits 1 / 2 of this switch remove the activation / weight traffic)
        rc = launch_splitk64_t<4, 4, 4, 4, 5, 3, false, 2>(sp, G, st);
        return rc == MI355_OK ? ns : rc;
    }
#endif
    if (wbits == 8) { SK_(8, 4); } else if (group_size == 128) { SK_(4, 4); } else if (group_size == 64) { SK_(4, 2); } else { SK_(4, 1); }
#undef SK_MB_
#undef SK_
    return rc == MI355_OK ? ns : rc;
}
