// Persistent weight-only GEMM for the smallest decode batches (M <= 8 rows), W4 / W8, gfx950.
//
// Why a second kernel: the staged-x kernel (gemm.hip) gives every 16-column tile to one wave for the whole K
// range and synchronises a block at every 128-k chunk.  At M <= 8 that leaves two losses the profiles show
// (profiles/r01_pmc_gemm_w4_m1.txt): 592 gate_up blocks on 256 CUs run as "3 blocks here, 2 there" (a 1.3x
// makespan), and the per-chunk barrier couples the waves of a block.  Here:
//   * one block of 16 waves per CU (grid ~ 256 blocks), each owning an equal share of the output tiles;
//   * the M activation rows of the block's k-range stay resident in LDS (<= 8 x 7 KB): no staging, no barrier
//     inside the loop, every wave runs free;
//   * inside the block the (tile, chunk) units are dealt to the waves as contiguous ranges of equal length
//     (stream-K inside the workgroup): a wave flushes its partial 16x16 tile to LDS when it crosses a tile
//     boundary, and after one barrier the tiles are summed in wave order (deterministic) and written with the
//     usual fused epilogue (bias / SiLU-gate / fp16 / fp32 / split-K slab).
// The arithmetic is the accumulator-side formulation of gemm.hip (biased codes 1024+u / 64+u into the MFMA,
// zero point and scale applied per quantisation group from the group sums of x); the group sums come from two
// extra "indicator" MFMAs per k-step — the matrix pipe is idle at M <= 8, VALU is the scarce resource.
#include "gemm_common.h"

namespace {

struct SmallMParams {
    GemmParams g;
    int GT;       // tile groups (grid.x)
    int xstride;  // bytes per x row in LDS (k-range * 2 + 16: rows land on consecutive 16-byte bank slots)
    int smax;     // partial-tile segments per tile
    int tiles_max;
};

template <int WBITS, int GS, int D, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_smallm_kernel(const SmallMParams sp) {
    const GemmParams& p = sp.g;
    constexpr int LPC  = WBITS / 4;
    constexpr int NSUB = (GS > 0) ? 4 / GS : 1;
    constexpr int SPG  = 4 / NSUB;
    constexpr bool GROUPED = GS > 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jj = lane & 15, q = lane >> 4;

    // ---- this block's share: tiles [t0, t1), chunks [c0, c0 + nch)
    const int t0 = (int)(((long)blockIdx.x * p.NT) / sp.GT), t1 = (int)(((long)(blockIdx.x + 1) * p.NT) / sp.GT);
    const int ntiles = t1 - t0;
    const int c0 = blockIdx.y * p.cps;
    const int nch = min(p.cps, p.KC - c0);
    const int U = ntiles * nch;                       // (tile, chunk) units of the block, tile-major
    const int Lw = (U + NW - 1) / NW;                 // units per wave
    const int u_begin = min(U, wave * Lw), u_end = min(U, u_begin + Lw);
    const int n_units = u_end - u_begin;

    // ---- LDS: x rows (+ one zero row), then the partial-tile slots
    char*  xs     = smem;
    const int xbytes = (p.M + 1) * sp.xstride;
    f32x4* pslots = reinterpret_cast<f32x4*>(smem + ((xbytes + 15) & ~15));

    {   // preload x[0..M) x [c0*128, c0*128 + nch*128) as 16-byte pieces; columns past K and the extra row are zero
        const int ppr = nch * 16;                     // pieces per row
        const int total = (p.M + 1) * ppr;
        for (int idx = tid; idx < total; idx += 64 * NW) {
            const int r = idx / ppr, pc = idx - r * ppr;
            const int k = c0 * 128 + pc * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r < p.M && k < p.K) v = *reinterpret_cast<const u32x4*>(p.x + (size_t)r * p.K + k);
            *reinterpret_cast<u32x4*>(xs + r * sp.xstride + pc * 16) = v;
        }
    }

    constexpr uint32_t FLAGS = 0x00020000u;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.qw, 0, p.qw_bytes, FLAGS);
    __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.meta, 0, p.meta_bytes, FLAGS);
    constexpr uint32_t OOBS = 0xFFFFFF00u;              // wave-uniform offset that is out of range for any image < 4 GiB

    u32x4    wr[D][LPC];
    u32x4    mr[D][NSUB];                         // meta {zneg, scale} of this lane's 4 output columns (C layout)
    const uint32_t lane16 = lane * 16u, q16 = q * 16u;
    const W4Consts w4c = w4_consts();

    // Cursors advance by constant strides (a handful of SALU per unit; an earlier version recomputed
    // tile * KC + chunk and a division per slot: ~150 scalar instructions per unit made the kernel scalar-bound).
    const int t_first = (nch > 0) ? u_begin / nch : 0, c_first = (nch > 0) ? u_begin - t_first * nch : 0;
    // NOTE: no `c ? a : b` between two captured variables inside the lambdas: InstCombine turns that into a load
    // through a selected pointer, the closure then lives in scratch and the offset lands in a VGPR (waterfall loop
    // + vmcnt(0) around every buffer load).  Masks / adds of single values keep everything in SGPRs.
    constexpr uint32_t w_step = LPC * 1024u;
    const uint32_t w_extra = (uint32_t)(p.KC - nch) * (LPC * 1024u);                              // tile wrap: + this
    const uint32_t m_step = (uint32_t)NSUB * (uint32_t)p.N_pad * 4u, m_extra = 64u - (uint32_t)nch * m_step; // mod 2^32
    uint32_t pf_w = ((uint32_t)(t0 + t_first) * (uint32_t)p.KC + (uint32_t)(c0 + c_first)) * (LPC * 1024u);
    uint32_t pf_m = ((uint32_t)(c0 + c_first) * NSUB * (uint32_t)p.N_pad + (uint32_t)(t0 + t_first) * 16u) * 4u;
    int pf_rem = n_units, pf_c = c_first;
    const uint32_t npad4 = (uint32_t)p.N_pad * 4u;
    auto load_w = [&](int d) {
        const bool ok = pf_rem > 0;
        const uint32_t okm = 0u - (uint32_t)ok;
        const uint32_t woff = (pf_w & okm) | (OOBS & ~okm);
#pragma unroll
        for (int lp = 0; lp < LPC; ++lp) wr[d][lp] = bload128<2 /*nt*/>(rw, lane16, woff + ((lp * 1024u) & okm));
        if (GROUPED) {
#pragma unroll
            for (int gi = 0; gi < NSUB; ++gi)
                mr[d][gi] = bload128<0>(rm, q16, ((pf_m + gi * npad4) & okm) | (OOBS & ~okm));
        }
        --pf_rem;
        const bool wrap = ++pf_c == nch;
        pf_c = wrap ? 0 : pf_c;
        const uint32_t wm = 0u - (uint32_t)wrap;
        pf_w += w_step + (wm & w_extra);
        pf_m += m_step + (wm & m_extra);
    };

    f16x8 ind0, ind1; // indicator operands: ones at the 1024-biased / 64-biased k-slots (see gemm.hip)
    {
        u32x4 i0 = {0x3C003C00u, 0u, 0x3C003C00u, 0u}, i1 = {0u, 0x3C003C00u, 0u, 0x3C003C00u};
        if (WBITS == 8) i0 = (u32x4){0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
        ind0 = __builtin_bit_cast(f16x8, i0); ind1 = __builtin_bit_cast(f16x8, i1);
    }

#pragma unroll
    for (int d = 0; d < D; ++d) load_w(d);
    __syncthreads(); // x resident

    // compute cursor
    int rem = n_units, cc = c_first, ct = t_first;
    int seg = (nch > 0 && Lw > 0) ? wave - (t_first * nch) / Lw : 0; // later tiles of this wave start inside its range: seg 0
    int xk = c_first * 256;
    const int row_real = (jj < p.M ? jj : p.M) * sp.xstride + q * 16; // lanes past M read the zero row
    const int row_zero = p.M * sp.xstride + q * 16, row_delta = row_real - row_zero;
    const int xs_off = 0, ps_off = (xbytes + 15) & ~15;               // byte offsets inside the dynamic LDS block
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};

    auto slot = [&](int d) {
        const bool valid = rem > 0;
        const int xbase = xs_off + row_zero + ((0 - (int)valid) & row_delta) + xk;  // padded slots multiply zeros
#pragma unroll
        for (int gi = 0; gi < NSUB; ++gi) {
            f32x4 ag = {0.f, 0.f, 0.f, 0.f}, x0a = ag, x1a = ag;
#pragma unroll
            for (int ss = 0; ss < SPG; ++ss) {
                const int s = gi * SPG + ss;
                const f16x8 b = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(smem + xbase + s * 64));
                x0a = mfma16x16x32(ind0, b, x0a);
                if (WBITS == 4) x1a = mfma16x16x32(ind1, b, x1a);
                f16x8 a;
                if (WBITS == 4) {
                    a = widen_w4(wr[d][0][s], w4c);
                } else {
                    const u32x4 w = wr[d][(s >> 1) % LPC];
                    a = widen_w8(w[(s & 1) * 2], w[(s & 1) * 2 + 1]);
                }
                ag = mfma16x16x32(a, b, ag);
            }
            const float XS = x0a[0] + x1a[0], XB = 960.f * x1a[0];
            if (GROUPED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f16x2 m = as_h2(mr[d][gi][r]);
                    const float t = __builtin_fmaf((float)m[0], XS, ag[r] + XB);
                    acc[r] = __builtin_fmaf((float)m[1], t, acc[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += __builtin_fmaf(-1152.f, XS, ag[r]);
            }
        }
        load_w(d);
        // tile boundary (or end of this wave's range): park the partial tile in LDS
        const bool tile_end = cc == nch - 1;
        if (valid && (tile_end || rem == 1)) {
            *reinterpret_cast<f32x4*>(smem + ps_off + ((ct * sp.smax + seg) * 64 + lane) * 16) = acc;
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        --rem;
        cc = tile_end ? 0 : cc + 1;
        xk = tile_end ? 0 : xk + 256;
        ct += tile_end ? 1 : 0;
        seg = tile_end ? 0 : seg;
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int it = 0; it < n_units; it += D) { // unguarded rounds (see gemm.hip): padded slots see OOB = 0 weights
#pragma unroll
        for (int d = 0; d < D; ++d) slot(d);
    }
    __syncthreads();

    // ---- sum the segments of each tile in wave order, epilogue
    for (int tl = wave; tl < ntiles; tl += NW) {
        const int first = (tl * nch) / Lw, last = ((tl + 1) * nch - 1) / Lw;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int sgm = 0; sgm <= last - first; ++sgm) v += pslots[((tl * sp.smax) + sgm) * 64 + lane];
        const int n0 = (t0 + tl) * 16 + q * 4;
        if (!GROUPED) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = __builtin_amdgcn_raw_buffer_load_b32(rm, (uint32_t)(n0 + r) * 4u, 0, 0);
                v[r] *= (float)as_h2(m)[1];
            }
        }
        if (jj < p.M) gemm_store(p, v, jj, n0, blockIdx.y);
    }
}

template <int WBITS, int GS, int D = 4>
int launch_t(const SmallMParams& sp, size_t lds, hipStream_t st) {
    constexpr int NW = 16;
    auto k = gemm_smallm_kernel<WBITS, GS, D, NW>;
    if (int e = raise_dynamic_lds((const void*)k, "gemm_smallm")) return e; // allow > 64 KiB of dynamic LDS
    hipLaunchKernelGGL(k, dim3(sp.GT, sp.g.nsplit), dim3(64 * NW), lds, st, sp);
    MI355_CHECK_LAUNCH("gemm_smallm_kernel");
    return MI355_OK;
}

} // namespace

// Plan + launch.  Returns MI355_ERR_UNSUPPORTED when the shape does not fit (caller falls back to gemm.hip).
// `want_partial`: write fp32 slabs (nsplit may be > 1); otherwise nsplit = 1 with the fused epilogue in p.mode.
extern "C" int mi355_gemm_smallm(const void* gp, int wbits, int group_size, int want_partial, int max_splits,
                                 mi355_stream_t stream) {
    GemmParams g = *reinterpret_cast<const GemmParams*>(gp);
    if (g.M > 8 || (wbits != 4 && wbits != 8)) return MI355_ERR_UNSUPPORTED;
    if (wbits == 4 && !(group_size == 128 || group_size == 64 || group_size == 32)) return MI355_ERR_UNSUPPORTED;
    if (wbits == 8 && !(group_size == 0 || group_size == 128)) return MI355_ERR_UNSUPPORTED;
    constexpr int NW = 16, CUS = 256;
    SmallMParams sp;
    // blocks ~ number of CUs: tile groups x k splits
    int nsplit = 1, GT = g.NT < CUS ? g.NT : CUS;
    if (want_partial) {
        const int tiles_per_blk = 8;
        GT = (g.NT + tiles_per_blk - 1) / tiles_per_blk;
        if (GT > CUS) GT = CUS;
        nsplit = CUS / GT;
        if (nsplit > max_splits) nsplit = max_splits;
        if (nsplit > g.KC / 2) nsplit = g.KC / 2 > 0 ? g.KC / 2 : 1;
        if (nsplit < 1) nsplit = 1;
    }
    g.cps = (g.KC + nsplit - 1) / nsplit;
    g.nsplit = (g.KC + g.cps - 1) / g.cps;
    sp.g = g; sp.GT = GT;
    sp.xstride = g.cps * 256 + 16;
    sp.tiles_max = (g.NT + GT - 1) / GT;
    // segments per tile <= ceil(nch / Lw) + 1 with Lw >= ntiles * nch / NW  ->  <= ceil(NW / ntiles_min) + 1
    const int ntiles_min = g.NT / GT > 0 ? g.NT / GT : 1;
    sp.smax = (NW + ntiles_min - 1) / ntiles_min + 1;
    if (sp.smax > NW) sp.smax = NW;
    const size_t lds = (((size_t)(g.M + 1) * sp.xstride + 15) & ~(size_t)15) + (size_t)sp.tiles_max * sp.smax * 1024;
    if (lds > 150 * 1024) return MI355_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int rc = MI355_ERR_UNSUPPORTED;
#ifdef MI355_TUNING
    if (wbits == 4 && group_size == 128 && TUNE(3) == 1) return (rc = launch_t<4, 4, 8>(sp, lds, st)) < 0 ? rc : g.nsplit;
#endif
    if (wbits == 4) rc = group_size == 128 ? launch_t<4, 4>(sp, lds, st) : group_size == 64 ? launch_t<4, 2>(sp, lds, st) : launch_t<4, 1>(sp, lds, st);
    else            rc = group_size == 0 ? launch_t<8, 0>(sp, lds, st) : launch_t<8, 4>(sp, lds, st);
    return rc < 0 ? rc : g.nsplit;
}
