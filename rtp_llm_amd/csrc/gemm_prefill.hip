// Weight-only dequant GEMM for large M (prefill chunks, M >= 128 rows), gfx950.
//
// Same contract and weight image as gemm.hip (reference slot: LinearBase.forward,
// rtp_llm/models_py/modules/factory/linear/linear_base.py:75-85; the reference notes that weight-only quantisation "may
// cause performance degradation for long sequences during the Prefill phase", docs/backend/quantization.md:3 -- the
// decode kernels of this library would re-read every weight once per 64 rows).  Here the problem is compute-shaped:
//   * block = 8 waves, tile = 128 rows x 256 columns; a wave owns 2 adjacent 16-column weight tiles and all 128 rows;
//   * per 128-k chunk a wave dequantises its 2 x 4 A-fragments ONCE (operand side, fp16) and uses each for the 8 row
//     blocks: 64 v_mfma_f32_16x16x32_f16 per 104 dequant VALU, so the matrix pipe, not the VALU, sets the pace; the second
//     wave of each SIMD dequantises while the first one multiplies;
//   * the activation chunk (128 rows x 128 k fp16 = 32 KiB) is staged once per block in LDS in the fragment-major,
//     XOR-swizzled image of gemm.hip (conflict-free ds_read_b128) and double-buffered; one B-fragment read feeds two MFMAs;
//   * weights stream from HBM/L2 once per 128 rows (the M/128 row blocks of a column block run back to back on different
//     CUs and meet in L2 / Infinity Cache), through a 2-deep register ring, buffer range checks instead of tail branches.
#include "gemm_common.h"

namespace {

// TPW = weight tiles per wave: 2 (tile 128 x 256, two blocks per CU) or 4 (128 x 512, one block per CU).  One B-fragment read from
// LDS feeds TPW MFMAs: at TPW = 2 the 8 waves of a block pull 256 KiB of fragments per 128-k chunk through the CU's LDS port for
// 512 MFMAs -- 2048 cycles of each, neither hidden behind the other; TPW = 4 halves the LDS traffic per MFMA.
template <int WBITS, int GS, int TPW>
__global__ __launch_bounds__(512, TPW == 2 ? 2 : 1) void gemm_prefill_kernel(const GemmParams p) {
    constexpr int BM = 128, MB = BM / 16, NW = 8;             // rows, row blocks, waves
    constexpr int LPC  = WBITS / 4;
    constexpr int NSUB = (GS > 0) ? 4 / GS : 1;
    constexpr int SPG  = 4 / NSUB;
    constexpr bool GROUPED = GS > 0;
    constexpr int XSLOTS = 256 * MB;                          // 16-byte pieces of an x chunk tile
    constexpr int UPT = XSLOTS / 512;                         // pieces per thread
    __shared__ u32x4 xs[2][XSLOTS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jj = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * BM;
    const int nt_base = (blockIdx.x * NW + wave) * TPW;
    const int KC = p.KC;

    constexpr uint32_t FLAGS = 0x00020000u;
    __amdgpu_buffer_rsrc_t rw[TPW], rm[TPW];
    bool tile_ok[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int nt = nt_base + t;
        tile_ok[t] = nt < p.NT;
        const char* wb = (const char*)p.qw + (size_t)nt * KC * (LPC * 1024);
        rw[t] = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, tile_ok[t] ? KC * LPC * 1024 : 0, FLAGS);
        const char* mb = (const char*)p.meta + (size_t)nt * 16 * 4;
        const int mbytes = GROUPED ? ((KC * NSUB - 1) * p.N_pad + 16) * 4 : 64;
        rm[t] = __builtin_amdgcn_make_buffer_rsrc((void*)mb, 0, tile_ok[t] ? mbytes : 0, FLAGS);
    }
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, FLAGS);

    // x staging: thread owns pieces pi = tid + 512 u: row j = pi >> 4 of the tile, 16-byte piece pp = pi & 15 of the chunk
    uint32_t xoff[UPT];
    int xslot[UPT];
    bool xrow_ok[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
        const int pi = tid + 512 * u, j = pi >> 4, pp = pi & 15;
        xrow_ok[u] = m0 + j < p.M;
        xoff[u]  = (uint32_t)(((size_t)(m0 + j) * p.K + pp * 8) * 2);
        xslot[u] = pp * (16 * MB) + (j ^ (pp & 3));
    }
    const uint32_t OOBX = 0x80000000u;
    auto load_x = [&](u32x4 (&xr)[UPT], int c) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const int pp = (tid + 512 * u) & 15;
            const bool ok = xrow_ok[u] && c < KC && c * 128 + pp * 8 < p.K;   // rows past M / columns past K read as zero
            xr[u] = bload128<0>(rx, ok ? xoff[u] : OOBX, (uint32_t)c * 256u);
        }
    };
    auto store_x = [&](const u32x4 (&xr)[UPT], int buf) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) xs[buf][xslot[u]] = xr[u];
    };

    const uint32_t lane16 = lane * 16u, jj4 = jj * 4u;
    u32x4    wr[2][TPW][LPC];
    uint32_t mr[2][TPW][NSUB];
    auto load_w = [&](int d, int c) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
#pragma unroll
            for (int lp = 0; lp < LPC; ++lp) wr[d][t][lp] = bload128<0>(rw[t], lane16, (uint32_t)(c * LPC + lp) * 1024u);
            if (GROUPED) {
#pragma unroll
                for (int gi = 0; gi < NSUB; ++gi)
                    mr[d][t][gi] = __builtin_amdgcn_raw_buffer_load_b32(rm[t], jj4, (uint32_t)(c * NSUB + gi) * (uint32_t)p.N_pad * 4u, 0);
            }
        }
    };
    uint32_t mch[TPW];
    if (!GROUPED) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) mch[t] = __builtin_amdgcn_raw_buffer_load_b32(rm[t], jj4, 0, 0);
    }

    f32x4 acc[TPW][MB];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const W4Consts w4c = w4_consts();
    const f16x2 c960 = {(f16)960.f, (f16)960.f};

    u32x4 xr[UPT];
    load_x(xr, 0);
    load_w(0, 0);
    store_x(xr, 0);
    load_x(xr, 1);
    load_w(1, 1);
    __syncthreads();

    auto dq = [&](int d, int t, int s) -> f16x8 {
        const uint32_t m = GROUPED ? mr[d][t][s / SPG] : mch[t];
        const f16x2 zn = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u));
        const f16x2 sc = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
        if (WBITS == 4) return dequant_w4_vc(wr[d][t][0][s], zn, zn + c960, sc, w4c);
        const u32x4 w = wr[d][t][(s >> 1) % LPC];   // per-channel int8: exact (u - z) operand, the column scale is applied in fp32 in the epilogue
        return dequant_w8<GROUPED>(w[(s & 1) * 2], w[(s & 1) * 2 + 1], zn, sc);
    };
    auto compute = [&](int d, int buf) {
        if constexpr (TPW == 2) {            // all 8 operands of the chunk first, then 64 MFMAs
            f16x8 a[TPW][4];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) a[t][s] = dq(d, t, s);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f16x8 b = __builtin_bit_cast(f16x8, xs[buf][(s * 4 + q) * (16 * MB) + mb * 16 + (jj ^ q)]);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) acc[t][mb] = mfma16x16x32(a[t][s], b, acc[t][mb]);
                }
        } else {                             // per k-step: TPW operands (16 registers), 8 fragment reads, 8 x TPW MFMAs
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 a[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) a[t] = dq(d, t, s);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const f16x8 b = __builtin_bit_cast(f16x8, xs[buf][(s * 4 + q) * (16 * MB) + mb * 16 + (jj ^ q)]);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) acc[t][mb] = mfma16x16x32(a[t], b, acc[t][mb]);
                }
            }
        }
    };

    for (int c = 0; c < KC; c += 2) {
        // chunk c (buffer 0, ring 0); x(c+1) is in registers, w(c+1) in ring 1
        compute(0, 0);
        store_x(xr, 1);
        load_x(xr, c + 2);
        load_w(0, c + 2);
        __syncthreads();
        if (c + 1 < KC) compute(1, 1);
        store_x(xr, 0);
        load_x(xr, c + 3);
        load_w(1, c + 3);
        __syncthreads();
    }

    // ---- epilogue: lane holds, for row m0 + mb*16 + jj, the 4 consecutive columns 16 nt + 4q + r
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (!tile_ok[t]) continue;
        const int n0 = (nt_base + t) * 16 + q * 4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f};
        if (!GROUPED) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = __builtin_amdgcn_raw_buffer_load_b32(rm[t], (uint32_t)(q * 4 + r) * 4u, 0, 0);
                sc[r] = (float)as_h2(m)[1];
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = m0 + mb * 16 + jj;
            if (m < p.M) gemm_store(p, acc[t][mb] * sc, m, n0, 0);
        }
    }
}

template <int WBITS, int GS>
int launch_prefill_t(const GemmParams& p, hipStream_t st) {
    // four tiles per wave (one block per CU) when its grid is one well-filled round of the 256 CUs or many rounds; in between
    // (257..~560 blocks: a second round that is mostly empty) and for small grids the two-tile shape with two blocks per CU wins
    // (profiles/r03_prefill_gemm_tiles_per_wave.txt: gate_up M = 4096 959 -> 1055 TFLOP/s, down 932 -> 1066, o 885 -> 1015;
    // qkv M = 4096 with 288 blocks 755 vs 667, gate_up M = 512 with 296 blocks 776 vs 701)
    const long b4 = (long)cdiv(p.NT, 32) * cdiv(p.M, 128);
    const bool wide = TUNE(3) == 4 || (TUNE(3) != 2 && ((b4 >= 140 && b4 <= 256) || b4 >= 560));
    dim3 grid(cdiv(p.NT, wide ? 32 : 16), cdiv(p.M, 128));
    if (wide) hipLaunchKernelGGL((gemm_prefill_kernel<WBITS, GS, 4>), grid, dim3(512), 0, st, p);
    else      hipLaunchKernelGGL((gemm_prefill_kernel<WBITS, GS, 2>), grid, dim3(512), 0, st, p);
    MI355_CHECK_LAUNCH("gemm_prefill_kernel");
    return MI355_OK;
}

} // namespace

// Direct-mode launch (fused epilogue in p.mode).  MI355_ERR_UNSUPPORTED: shape / format not covered, caller falls back.
extern "C" int mi355_gemm_prefill(const void* gp, int wbits, int group_size, mi355_stream_t stream) {
    const GemmParams& g = *reinterpret_cast<const GemmParams*>(gp);
    if (g.M < 128 || g.mode == MODE_PARTIAL || g.mode == MODE_F32) return MI355_ERR_UNSUPPORTED;
    if ((uint64_t)g.M * g.K * 2 >= 0x7FFFFFF0ull) return MI355_ERR_UNSUPPORTED;       // x image must stay below the OOB offset
    hipStream_t st = (hipStream_t)stream;
    if (wbits == 4) {
        if (group_size == 128) return launch_prefill_t<4, 4>(g, st);
        if (group_size == 64) return launch_prefill_t<4, 2>(g, st);
        if (group_size == 32) return launch_prefill_t<4, 1>(g, st);
    }
    if (wbits == 8 && group_size == 0) return launch_prefill_t<8, 0>(g, st);
    return MI355_ERR_UNSUPPORTED;
}
