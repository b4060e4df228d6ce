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
#include "common.h"
#include "internal.h"

namespace {

struct GemmParams {
    const f16*      x;
    const void*     qw;
    const uint32_t* meta;
    const f16*      bias;
    void*           y;        // direct mode output
    float*          partials; // partial mode output
    int M, K;                 // logical K (row stride of x)
    int N, N_pad, NT, KC;     // NT = N_pad/16, KC = K_pad/128
    int nsplit, cps;          // chunks per split
    int mode;                 // 0 partial slabs, 1 fp16, 2 fp16 silu-mul, 3 fp32
    int ldy;
    uint32_t qw_bytes, meta_bytes, x_bytes;
};

enum { MODE_PARTIAL = 0, MODE_F16 = 1, MODE_SILU = 2, MODE_F32 = 3 };

template <int AUX>
__device__ __forceinline__ u32x4 bload128(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
}

template <int WBITS, int MB, int NBW, int GS, int D>
__global__ __launch_bounds__(256) void gemm_wq_kernel(const GemmParams p) {
    constexpr int LPC   = WBITS / 4;
    constexpr int GSD   = (GS > 0) ? GS : 1; // divisor-safe
    constexpr int NMETA = (GS > 0) ? 4 / GSD : 0;
    constexpr int XSLOTS = 256 * MB; // 16-byte slots per x chunk tile
    __shared__ u32x4 xs[2][XSLOTS];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jj = lane & 15, q = lane >> 4;

    const int nt_base = (blockIdx.x * 4 + wave) * NBW;
    const int c_begin = blockIdx.y * p.cps;
    const int n_ch    = min(p.cps, p.KC - c_begin);

    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.qw, 0, p.qw_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.meta, 0, p.meta_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

    const uint32_t OOB = 0xFFFFFFF0u;

    // per-tile byte offsets (chunk 0 of this split), OOB when the tile does not exist
    uint32_t woff[NBW];
    uint32_t moff[NBW];
    bool     tile_ok[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int nt = nt_base + nb;
        tile_ok[nb]  = nt < p.NT;
        woff[nb] = tile_ok[nb] ? (uint32_t)(((uint32_t)nt * p.KC + c_begin) * LPC * 1024u + lane * 16u) : OOB;
        moff[nb] = tile_ok[nb] ? (uint32_t)((nt * 16 + jj) * 4) : OOB;
    }

    u32x4    wr[D][NBW][LPC];
    uint32_t mr[D][NBW][NMETA > 0 ? NMETA : 1];
    f32x4    acc[NBW][MB];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto load_w = [&](int d, int ci) {
        const bool ok = ci < n_ch;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const uint32_t base = (ok && tile_ok[nb]) ? woff[nb] + (uint32_t)ci * (LPC * 1024u) : OOB;
#pragma unroll
            for (int lp = 0; lp < LPC; ++lp) wr[d][nb][lp] = bload128<2 /*nt*/>(rw, base + lp * 1024u);
            if (NMETA > 0) {
#pragma unroll
                for (int gi = 0; gi < NMETA; ++gi) {
                    const int g = (c_begin + ci) * NMETA + gi;
                    const uint32_t mo = (ok && tile_ok[nb]) ? moff[nb] + (uint32_t)g * (uint32_t)p.N_pad * 4u : OOB;
                    mr[d][nb][gi] = __builtin_amdgcn_raw_buffer_load_b32(rm, mo, 0, 0);
                }
            }
        }
    };

    // x staging: thread owns MB 16-byte pieces of the [16*MB rows][128 k] chunk tile
    uint32_t xoff[MB];
    int      xslot[MB];
#pragma unroll
    for (int u = 0; u < MB; ++u) {
        const int pi = tid + 256 * u;
        const int j = pi >> 4, pp = pi & 15;
        xoff[u]  = (j < p.M) ? (uint32_t)((j * p.K + c_begin * 128 + pp * 8) * 2) : OOB;
        xslot[u] = pp * (16 * MB) + (j ^ (pp & 3));
    }
    const int kpiece0 = c_begin * 128; // k of piece 0 in chunk 0
    u32x4 xr[MB];
    auto load_x = [&](int ci) {
#pragma unroll
        for (int u = 0; u < MB; ++u) {
            const int pp = (tid + 256 * u) & 15;
            const bool ok = (ci < n_ch) && (kpiece0 + ci * 128 + pp * 8 < p.K) && (xoff[u] != OOB);
            xr[u] = bload128<0>(rx, ok ? xoff[u] + (uint32_t)ci * 256u : OOB);
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int u = 0; u < MB; ++u) xs[buf][xslot[u]] = xr[u];
    };

    // per-channel mode: zero term is constant along K
    uint32_t mch[NBW];
    if (GS == 0 && WBITS != 16) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) mch[nb] = __builtin_amdgcn_raw_buffer_load_b32(rm, moff[nb], 0, 0);
    }

    // prologue
#pragma unroll
    for (int d = 0; d < D; ++d) load_w(d, d);
    load_x(0);
    store_x(0);
    __syncthreads();

    auto compute = [&](int d, int buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f16x8 b[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const u32x4 v = xs[buf][(s * 4 + q) * (16 * MB) + mb * 16 + (jj ^ q)];
                b[mb] = __builtin_bit_cast(f16x8, v);
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                f16x8 a;
                if (WBITS == 16) {
                    a = __builtin_bit_cast(f16x8, wr[d][nb][s % LPC]);
                } else {
                    const uint32_t m = (GS > 0) ? mr[d][nb][(GS > 0) ? s / GSD : 0] : mch[nb];
                    const f16x2 zneg2 = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u));
                    const f16x2 s2    = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
                    if (WBITS == 4) {
                        a = dequant_w4(wr[d][nb][0][s], zneg2, s2);
                    } else {
                        const u32x4 w = wr[d][nb][(s >> 1) % LPC];
                        a = dequant_w8<(GS > 0)>(w[(s & 1) * 2], w[(s & 1) * 2 + 1], zneg2, s2);
                    }
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = mfma16x16x32(a, b[mb], acc[nb][mb]);
            }
        }
    };

    for (int it = 0; it < n_ch; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int ci = it + d;
            if (ci < n_ch) {
                load_x(ci + 1);          // OOB -> zeros past the end
                compute(d, d & 1);       // D is even, so (it + d) & 1 == d & 1
                load_w(d, ci + D);
                store_x((d + 1) & 1);
                __syncthreads();
            }
        }
    }

    // ------------------------------------------------------------ epilogue
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        if (!tile_ok[nb]) continue;
        const int n0 = (nt_base + nb) * 16 + q * 4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f};
        if (GS == 0 && WBITS != 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = __builtin_amdgcn_raw_buffer_load_b32(rm, (uint32_t)(n0 + r) * 4u, 0, 0);
                sc[r] = (float)as_h2(m)[1];
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = mb * 16 + jj;
            if (m >= p.M) continue;
            f32x4 v = acc[nb][mb] * sc;
            if (p.mode == MODE_PARTIAL) {
                float* dst = p.partials + ((size_t)blockIdx.y * p.M + m) * p.N_pad + n0;
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
                if (n0 >= p.N) continue;
                if (p.bias) {
                    const f16x4 bv = *reinterpret_cast<const f16x4*>(p.bias + n0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
                }
                if (p.mode == MODE_F32) {
                    *reinterpret_cast<f32x4*>((float*)p.y + (size_t)m * p.ldy + n0) = v;
                } else if (p.mode == MODE_F16) {
                    f16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (f16)v[r];
                    *reinterpret_cast<f16x4*>((f16*)p.y + (size_t)m * p.ldy + n0) = o;
                } else { // MODE_SILU: (gate, up) interleaved; round GEMM output to fp16 first
                    f16x2 o;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float g = (float)(f16)v[2 * t], u = (float)(f16)v[2 * t + 1];
                        o[t] = (f16)((g / (1.f + __expf(-g))) * u);
                    }
                    *reinterpret_cast<f16x2*>((f16*)p.y + (size_t)m * p.ldy + (n0 >> 1)) = o;
                }
            }
        }
    }
}

// Sum split-K slabs (+bias) and apply the epilogue.  One thread per 4 columns.
__global__ __launch_bounds__(256) void reduce_epilogue_kernel(const float* __restrict__ partials, int nsplit,
                                                              int M, int N, int N_pad, const f16* __restrict__ bias,
                                                              void* y, int ldy, int mode) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n4  = N_pad >> 2;
    if (idx >= M * n4) return;
    const int m = idx / n4, n0 = (idx - m * n4) * 4;
    if (n0 >= N) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s)
        v += *reinterpret_cast<const f32x4*>(partials + ((size_t)s * M + m) * N_pad + n0);
    if (bias) {
        const f16x4 bv = *reinterpret_cast<const f16x4*>(bias + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
    }
    if (mode == MODE_F32) {
        *reinterpret_cast<f32x4*>((float*)y + (size_t)m * ldy + n0) = v;
    } else if (mode == MODE_F16) {
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (f16)v[r];
        *reinterpret_cast<f16x4*>((f16*)y + (size_t)m * ldy + n0) = o;
    } else {
        f16x2 o;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float g = (float)(f16)v[2 * t], u = (float)(f16)v[2 * t + 1];
            o[t] = (f16)((g / (1.f + __expf(-g))) * u);
        }
        *reinterpret_cast<f16x2*>((f16*)y + (size_t)m * ldy + (n0 >> 1)) = o;
    }
}

// ------------------------------------------------------------------ dispatch
template <int WBITS, int GS, int D>
int launch_gemm_t(const GemmParams& p, int MB, int NBW, hipStream_t st) {
    dim3 grid(cdiv(p.NT, 4 * NBW), p.nsplit), block(256);
#define L_(mb, nbw)                                                                           \
    hipLaunchKernelGGL((gemm_wq_kernel<WBITS, mb, nbw, GS, D>), grid, block, 0, st, p);        \
    break;
    if (NBW == 1) {
        switch (MB) { case 1: L_(1, 1) case 2: L_(2, 1) case 3: L_(3, 1) default: L_(4, 1) }
    } else {
        switch (MB) { case 1: L_(1, 2) case 2: L_(2, 2) case 3: L_(3, 2) default: L_(4, 2) }
    }
#undef L_
    MI355_CHECK_LAUNCH("gemm_wq_kernel");
    return MI355_OK;
}

int launch_gemm(const GemmParams& p, int wbits, int group_size, int MB, int NBW, hipStream_t st) {
    if (wbits == 16) return launch_gemm_t<16, 0, 2>(p, MB, NBW, st);
    if (wbits == 4) {
        if (group_size == 128) return launch_gemm_t<4, 4, 4>(p, MB, NBW, st);
        if (group_size == 64) return launch_gemm_t<4, 2, 4>(p, MB, NBW, st);
        if (group_size == 32) return launch_gemm_t<4, 1, 4>(p, MB, NBW, st);
    }
    if (wbits == 8) {
        if (group_size == 0) return launch_gemm_t<8, 0, 4>(p, MB, NBW, st);
        if (group_size == 128) return launch_gemm_t<8, 4, 4>(p, MB, NBW, st);
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
}

} // namespace

// Split-K plan shared with the engine: aim at ~2 blocks per CU, >= 2 chunks per split.
extern "C" int mi355_gemm_plan(int M, const mi355_weight_t* w, int max_splits, int* nbw_out, int* cps_out) {
    const int NT = w->N_pad / 16, KC = w->K_pad / 128;
    int NBW = (NT >= 4096) ? 2 : 1;
    const int blocks_n = cdiv(NT, 4 * NBW);
    int target = 512;
    int nsplit = target / blocks_n;
    if (nsplit < 1) nsplit = 1;
    // large M: slab traffic grows with M*nsplit, keep it below ~half the weight bytes
    const double wbytes = (double)w->K_pad * w->N_pad * w->wbits / 8.0;
    while (nsplit > 1 && (double)nsplit * (M < 16 ? 16 : M) * w->N_pad * 8.0 > wbytes) --nsplit;
    if (nsplit > max_splits) nsplit = max_splits;
    if (nsplit > KC / 2) nsplit = KC / 2 > 0 ? KC / 2 : 1;
    int cps = cdiv(KC, nsplit);
    nsplit  = cdiv(KC, cps);
    *nbw_out = NBW; *cps_out = cps;
    return nsplit;
}

extern "C" size_t mi355_linear_workspace_bytes(int32_t M, const mi355_weight_t* w) {
    if (!w || M <= 0) return 0;
    const int Mc = M > 64 ? 64 : M;
    int nbw, cps;
    const int ns = mi355_gemm_plan(Mc, w, 64, &nbw, &cps);
    return ns > 1 ? (size_t)ns * Mc * w->N_pad * sizeof(float) : 0;
}

extern "C" int mi355_linear_partial(const void* x, int32_t M, const mi355_weight_t* w, float* partials,
                                    int32_t max_splits, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x && partials && M > 0 && M <= 64 && max_splits >= 1, "linear_partial: bad args (M=%d)", M);
    GemmParams p; fill_params(p, x, M, w);
    int nbw, cps;
    p.nsplit = mi355_gemm_plan(M, w, max_splits, &nbw, &cps);
    p.cps = cps; p.mode = MODE_PARTIAL; p.partials = partials;
    if (int e = launch_gemm(p, w->wbits, w->group_size, cdiv(M, 16), nbw, (hipStream_t)stream)) return e;
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
    for (int m0 = 0; m0 < M; m0 += 64) {
        const int Mc = (M - m0) > 64 ? 64 : (M - m0);
        GemmParams p; fill_params(p, (const f16*)x + (size_t)m0 * w->K, Mc, w);
        int nbw, cps;
        int ns = mi355_gemm_plan(Mc, w, 64, &nbw, &cps);
        if (ns > 1 && (size_t)ns * Mc * w->N_pad * sizeof(float) > workspace_bytes) {
            mi355_set_error("linear_forward: workspace %zu too small", workspace_bytes);
            return MI355_ERR_WORKSPACE;
        }
        void* yc = (char*)y + (size_t)m0 * ldy * ysz;
        p.nsplit = ns; p.cps = cps; p.ldy = ldy;
        if (ns == 1) {
            p.mode = mode; p.bias = (const f16*)bias; p.y = yc;
            if (int e = launch_gemm(p, w->wbits, w->group_size, cdiv(Mc, 16), nbw, st)) return e;
        } else {
            p.mode = MODE_PARTIAL; p.partials = (float*)workspace;
            if (int e = launch_gemm(p, w->wbits, w->group_size, cdiv(Mc, 16), nbw, st)) return e;
            const int total = Mc * (w->N_pad / 4);
            hipLaunchKernelGGL(reduce_epilogue_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)workspace,
                               ns, Mc, w->N, w->N_pad, (const f16*)bias, yc, ldy, mode);
            MI355_CHECK_LAUNCH("reduce_epilogue_kernel");
        }
    }
    return MI355_OK;
}

// Engine-internal: direct (nsplit = 1) GEMM with fused epilogue, no workspace.
extern "C" int mi355_linear_direct(const void* x, int32_t M, const mi355_weight_t* w, const void* bias, void* y,
                                   int32_t epilogue, mi355_stream_t stream) {
    if (int e = check_weight(w)) return e;
    MI355_CHECK_ARG(x && y && M > 0 && M <= 64, "linear_direct: bad args");
    const int mode = (epilogue & MI355_EPI_OUT_F32) ? MODE_F32 : (epilogue & MI355_EPI_SILU_MUL) ? MODE_SILU : MODE_F16;
    GemmParams p; fill_params(p, x, M, w);
    int nbw, cps;
    mi355_gemm_plan(M, w, 1, &nbw, &cps);
    p.nsplit = 1; p.cps = p.KC; p.mode = mode; p.bias = (const f16*)bias; p.y = y;
    p.ldy = (mode == MODE_SILU) ? w->N / 2 : w->N;
    return launch_gemm(p, w->wbits, w->group_size, cdiv(M, 16), nbw, (hipStream_t)stream);
}
