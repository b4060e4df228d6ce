// Row-wise fused epilogue kernels of the decode layer: (split-K reduce +) bias +
// residual add + RMSNorm, SiLU-gate, embedding gather, greedy argmax.  gfx950.
//
// These are launch/latency-bound (M <= 64 rows of a few KB); the design goal is
// to need as few launches per layer as possible, so the split-K reduction of the
// preceding GEMM and the residual add are folded into the norm kernel.
#include "common.h"
#include "internal.h"

namespace {

// ---------------------------------------------------------------- RMSNorm
// Reference numerics (rtp_llm/models_py/modules/base/common/norm.py:83-92):
//   var = mean(x_f32^2); xn = (x_f32 * rsqrt(var + eps)).to(fp16); y = weight * xn
// One block (256 threads) per row; each thread owns 8-element vectors.
template <int VPT, bool BF> // vectors (of 8 elements) per thread; 512 threads per row; BF: the rows / weight / bias are bf16
__global__ __launch_bounds__(512) void add_rmsnorm_kernel(const f16* __restrict__ x, const float* __restrict__ partials,
                                                          int nsplit, int ld, int M, const f16* __restrict__ bias,
                                                          const f16* __restrict__ res_in, f16* __restrict__ res_out,
                                                          const f16* __restrict__ weight, float eps, int H,
                                                          f16* __restrict__ y, int y_img_mblk, const mi355_touch_t tc) {
    constexpr int NTH = 512;
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    if (row >= M) {   // spare blocks (internal.h, mi355_touch_t): unit u of the next launch, from a block on the XCD that will run it
        const int q = row - M, u = (row & 7) + 8 * (q >> 3);
        if (u < tc.n_units) {
            for (int i = 0; i < tc.delay; ++i) __builtin_amdgcn_s_sleep(8);
            const uint32_t* qw = (const uint32_t*)tc.qw; const uint32_t* meta = (const uint32_t*)tc.meta;
            const int t0 = (u / tc.hh) * 2 * tc.hh + u % tc.hh;
            const uint32_t lines = tc.run_bytes >> 7, run_dw = tc.run_bytes >> 2;
            uint32_t acc = 0;
            for (uint32_t l = tid; l < 2 * lines; l += NTH) {
                const uint32_t r = l >= lines ? 1u : 0u;
                acc ^= qw[(size_t)(t0 + r * tc.hh) * run_dw + (l - r * lines) * 32];
            }
            for (uint32_t g = tid; g < 2 * tc.meta_groups; g += NTH) {
                const uint32_t r = g >= tc.meta_groups ? 1u : 0u;
                acc ^= meta[(size_t)(g - r * tc.meta_groups) * tc.meta_stride + (t0 + r * tc.hh) * 16];
            }
            if (acc == 0x9E3779B9u) *(uint32_t*)tc.sink = acc;   // practically never: keeps the loads alive
        }
        return;
    }
    const int nvec = H >> 3;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    float v[VPT][8];
    u32x4 rin[VPT], bin[VPT], win[VPT];
    // everything that does not depend on the slab sum is requested first (one memory round trip for all of it)
#pragma unroll
    for (int t = 0; t < VPT; ++t) {
        const int vi = tid + t * NTH;
        const int c0 = (vi < nvec ? vi : 0) * 8;
        rin[t] = res_in ? *reinterpret_cast<const u32x4*>(res_in + (size_t)row * H + c0) : zero4;
        bin[t] = bias ? *reinterpret_cast<const u32x4*>(bias + c0) : zero4;
        win[t] = (y && weight) ? *reinterpret_cast<const u32x4*>(weight + c0) : zero4;
    }
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < VPT; ++t) {
        const int vi = tid + t * NTH;
        if (vi < nvec) {
            const int c0 = vi * 8;
            if (partials) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[t][e] = 0.f;
                // slabs are summed in index order (deterministic)
                const size_t sstride = (size_t)M * ld;
                const float* src0 = partials + (size_t)row * ld + c0;
                int s = 0;
                if (nsplit > 8) { // 9..16 slabs (down_proj at 64 rows: 15): all of them in ONE round trip, summed in index order as below
                    f32x4 a[16], b[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const bool ok = u < nsplit;
                        const float* sp = src0 + (ok ? u : 0) * sstride;
                        a[u] = *reinterpret_cast<const f32x4*>(sp);
                        b[u] = *reinterpret_cast<const f32x4*>(sp + 4);
                        if (!ok) { a[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; b[u] = a[u]; }
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[t][e] += a[u][e]; v[t][4 + e] += b[u][e]; }
                    s = 16;
                }
                for (; s + 8 <= nsplit; s += 8) {
                    f32x4 a[8], b[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        a[u] = *reinterpret_cast<const f32x4*>(src0 + (s + u) * sstride);
                        b[u] = *reinterpret_cast<const f32x4*>(src0 + (s + u) * sstride + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[t][e] += a[u][e]; v[t][4 + e] += b[u][e]; }
                }
                if (s < nsplit) { // tail: predicated loads, still one round trip
                    f32x4 a[8], b[8];
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        const bool ok = s + u < nsplit;
                        const float* sp = src0 + (ok ? (s + u) : s) * sstride;
                        a[u] = *reinterpret_cast<const f32x4*>(sp);
                        b[u] = *reinterpret_cast<const f32x4*>(sp + 4);
                        if (!ok) { a[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; b[u] = a[u]; }
                    }
#pragma unroll
                    for (int u = 0; u < 7; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[t][e] += a[u][e]; v[t][4 + e] += b[u][e]; }
                }
            } else {
                act_unpack8<BF>(*reinterpret_cast<const u32x4*>(x + (size_t)row * H + c0), v[t]);
            }
            float aux[8];
            if (bias) {
                act_unpack8<BF>(bin[t], aux);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[t][e] += aux[e];
            }
            if (partials) { // the GEMM output is a 16-bit tensor in the reference: round once
#pragma unroll
                for (int e = 0; e < 8; ++e) v[t][e] = act_round<BF>(v[t][e]);
            }
            if (res_in) {
                act_unpack8<BF>(rin[t], aux);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[t][e] = act_round<BF>(v[t][e] + aux[e]);
            }
            if (res_out) *reinterpret_cast<u32x4*>(res_out + (size_t)row * H + c0) = act_pack8<BF>(v[t]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[t][e] * v[t][e];
        }
    }
    __shared__ float red[NTH / 64];
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NTH / 64; ++i) tot += red[i];
    const float rs = rsqrtf(tot / (float)H + eps);
    if (!y) return;
#pragma unroll
    for (int t = 0; t < VPT; ++t) {
        const int vi = tid + t * NTH;
        if (vi < nvec) {
            const int c0 = vi * 8;
            float wv[8], o[8];
            act_unpack8<BF>(win[t], wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = wv[e] * act_round<BF>(v[t][e] * rs);   // product of two 16-bit tensors: rounded once by the pack
            // y_img_mblk > 0: y is an activation image (common.h act_img_index): the same 16 bytes at another address, always fp16
            if (y_img_mblk > 0) *reinterpret_cast<u32x4*>(y + act_img_index(row, c0, y_img_mblk)) = img_pack8<BF>(o);
            else                *reinterpret_cast<u32x4*>(y + (size_t)row * H + c0) = act_pack8<BF>(o);
        }
    }
}

template <bool BF>
__global__ __launch_bounds__(256) void silu_mul_kernel(const f16* __restrict__ gu, int M, int I, f16* __restrict__ out) {
    const int nvec = I >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * nvec) return;
    const int m = idx / nvec, c0 = (idx - m * nvec) * 8;
    float g[8], u[8], o[8];
    act_unpack8<BF>(*reinterpret_cast<const u32x4*>(gu + (size_t)m * 2 * I + c0), g);
    act_unpack8<BF>(*reinterpret_cast<const u32x4*>(gu + (size_t)m * 2 * I + I + c0), u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (g[e] / (1.f + __expf(-g[e]))) * u[e];
    *reinterpret_cast<u32x4*>(out + (size_t)m * I + c0) = act_pack8<BF>(o);
}

__global__ __launch_bounds__(256) void embedding_kernel(const int32_t* __restrict__ ids, int T, const f16* __restrict__ table,
                                                        int H, int vocab, f16* __restrict__ out) {
    const int nvec = H >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * nvec) return;
    const int t = idx / nvec, c0 = (idx - t * nvec) * 8;
    int id = ids[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    *reinterpret_cast<f16x8*>(out + (size_t)t * H + c0) = *reinterpret_cast<const f16x8*>(table + (size_t)id * H + c0);
}

// ---------------------------------------------------------------- weight prefetch
// Touch `bytes` of read-only data so that they sit in the Infinity Cache / L2 when the next GEMM asks for them: issued on a
// side stream while a latency-bound all-reduce kernel (<= 64 blocks) occupies the main one (tensor-parallel step).
__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* __restrict__ p, size_t nvec, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const u32x4 v = p[i];
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x9E3779B9u && sink) *sink = acc;   // practically never: keeps the loads alive without a store per thread
}

// ---------------------------------------------------------------- argmax
// Stage 1: grid (B, 64): each block scans a contiguous 1/64 of the row.
// Stage 2: one wave per row picks the best of the 64 candidates.
// Ties resolve to the lowest index (torch.argmax on this platform returns the first max).
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ __launch_bounds__(256) void argmax_stage1(const float* __restrict__ logits, int V, int ld, float* __restrict__ cand_v,
                                                     int* __restrict__ cand_i) {
    const int b = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
    const int per = ((V + nparts - 1) / nparts + 3) & ~3;
    const int lo = part * per, hi = min(V, lo + per);
    const float* row = logits + (size_t)b * ld;
    float bv = -INFINITY; int bi = 0x7FFFFFFF;
    for (int i = lo + threadIdx.x * 4; i < hi; i += 256 * 4) {
        if (i + 3 < hi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (v[e] > bv) { bv = v[e]; bi = i + e; }
        } else {
            for (int e = 0; e < 4 && i + e < hi; ++e) if (row[i + e] > bv) { bv = row[i + e]; bi = i + e; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
        argmax_combine(bv, bi, ov, oi);
    }
    __shared__ float sv[4]; __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) argmax_combine(bv, bi, sv[w], si[w]);
        cand_v[b * nparts + part] = bv; cand_i[b * nparts + part] = bi;
    }
}

__global__ __launch_bounds__(64) void argmax_stage2(const float* __restrict__ cand_v, const int* __restrict__ cand_i, int nparts,
                                                    int32_t* __restrict__ ids, int32_t* __restrict__ positions) {
    const int b = blockIdx.x, l = threadIdx.x;
    float bv = l < nparts ? cand_v[b * nparts + l] : -INFINITY;
    int bi = l < nparts ? cand_i[b * nparts + l] : 0x7FFFFFFF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
        argmax_combine(bv, bi, ov, oi);
    }
    if (l == 0) {
        ids[b] = bi;
        if (positions) positions[b] += 1;
    }
}

// Vocab-split greedy over an external transport: stage 2 leaves one (max, GLOBAL index) pair per row, the pairs of all
// ranks are all-gathered, argmax_pick takes the best per row (ties: lowest global index, as a full-row argmax would).
struct ArgPair { float v; int i; };

__global__ __launch_bounds__(64) void argmax_pair(const float* __restrict__ cand_v, const int* __restrict__ cand_i, int nparts,
                                                  int vocab_offset, ArgPair* __restrict__ out) {
    const int b = blockIdx.x, l = threadIdx.x;
    float bv = l < nparts ? cand_v[b * nparts + l] : -INFINITY;
    int bi = l < nparts ? cand_i[b * nparts + l] : 0x7FFFFFFF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
        argmax_combine(bv, bi, ov, oi);
    }
    if (l == 0) out[b] = ArgPair{bv, bi == 0x7FFFFFFF ? bi : bi + vocab_offset};
}

__global__ __launch_bounds__(64) void argmax_pick(const ArgPair* __restrict__ pairs, int world, int B, int32_t* __restrict__ ids,
                                                  int32_t* __restrict__ positions) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float bv = -INFINITY; int bi = 0x7FFFFFFF;
    for (int r = 0; r < world; ++r) { const ArgPair p = pairs[(size_t)r * B + b]; argmax_combine(bv, bi, p.v, p.i); }
    ids[b] = bi;
    if (positions) positions[b] += 1;
}

} // namespace

static int add_rmsnorm_launch(const void* x_f16, const float* partials, int32_t nsplit, int32_t ld, const void* bias,
                              const void* residual_in, void* residual_out, const void* weight, float eps, int32_t M,
                              int32_t H, void* y, int y_img_mblk, int32_t act_dtype, mi355_stream_t stream, const mi355_touch_t* touch = nullptr) {
    MI355_CHECK_ARG((x_f16 != nullptr) != (partials != nullptr), "add_rmsnorm: exactly one of x_f16 / partials");
    MI355_CHECK_ARG(M > 0 && H > 0 && H % 8 == 0 && H <= 8 * 512 * 2, "add_rmsnorm: M=%d H=%d (H %% 8 == 0, H <= 8192)", M, H);
    MI355_CHECK_ARG(!partials || (nsplit >= 1 && ld >= H && ld % 4 == 0), "add_rmsnorm: nsplit=%d ld=%d", nsplit, ld);
    MI355_CHECK_ARG(!y || weight, "add_rmsnorm: weight required");
    MI355_CHECK_ARG(act_dtype == MI355_ACT_F16 || act_dtype == MI355_ACT_BF16, "add_rmsnorm: act_dtype=%d", act_dtype);
    hipStream_t st = (hipStream_t)stream;
    const int vpt = cdiv(H / 8, 512);
    mi355_touch_t tc = {};
    int spare = 0;                                   // blocks past the rows: 8 per 8 units (see the kernel), only while the launch stays inside one round of the CUs
    static const int n_cus = [] {                    // CUs of the device (MI355X in SPX mode: 256); the touch relies on block b running on XCD b % 8
        int dev = 0, n = 0;                          // (observed there, speed only): other CU counts / partition modes get no spare blocks
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        (void)hipGetLastError();
        return n;
    }();
    if (touch && touch->qw && touch->n_units > 0 && touch->hh > 0 && n_cus == 256 && M + cdiv(touch->n_units, 8) * 8 <= n_cus) { tc = *touch; spare = cdiv(touch->n_units, 8) * 8; if (TUNE(3) > 0) tc.delay = TUNE(3); }
#define L_(V, B)                                                                                                        \
    hipLaunchKernelGGL((add_rmsnorm_kernel<V, B>), dim3(M + spare), dim3(512), 0, st, (const f16*)x_f16, partials, nsplit, ld, M, \
                       (const f16*)bias, (const f16*)residual_in, (f16*)residual_out, (const f16*)weight, eps, H, (f16*)y, y_img_mblk, tc)
    if (act_dtype == MI355_ACT_BF16) { if (vpt <= 1) L_(1, true); else L_(2, true); }
    else                             { if (vpt <= 1) L_(1, false); else L_(2, false); }
#undef L_
    MI355_CHECK_LAUNCH("add_rmsnorm_kernel");
    return MI355_OK;
}

extern "C" int mi355_add_rmsnorm_dt(const void* x_f16, const float* partials, int32_t nsplit, int32_t ld, const void* bias,
                                    const void* residual_in, void* residual_out, const void* weight, float eps, int32_t M,
                                    int32_t H, void* y, int32_t act_dtype, mi355_stream_t stream) {
    return add_rmsnorm_launch(x_f16, partials, nsplit, ld, bias, residual_in, residual_out, weight, eps, M, H, y, 0, act_dtype, stream);
}

// the same with y written as an activation image (mi355_act_image_*) for the full-K launches of a 5-64-row step
extern "C" int mi355_add_rmsnorm_img(const void* x_f16, const float* partials, int32_t nsplit, int32_t ld, const void* bias,
                                     const void* residual_in, void* residual_out, const void* weight, float eps, int32_t M,
                                     int32_t H, void* y_img, int32_t act_dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(y_img && M <= 64 && H % 32 == 0, "add_rmsnorm_img: M=%d (<= 64) H=%d (%% 32 == 0), y_img required", M, H);
    return add_rmsnorm_launch(x_f16, partials, nsplit, ld, bias, residual_in, residual_out, weight, eps, M, H, y_img, cdiv(M, 16), act_dtype, stream);
}

// the same launch with its spare blocks touching the weights of the launch that follows (internal.h: mi355_touch_t)
extern "C" int mi355_add_rmsnorm_img_touch(const void* x_f16, const float* partials, int32_t nsplit, int32_t ld, const void* bias,
                                           const void* residual_in, void* residual_out, const void* weight, float eps, int32_t M,
                                           int32_t H, void* y_img, int32_t act_dtype, const mi355_touch_t* touch, mi355_stream_t stream) {
    MI355_CHECK_ARG(y_img && M <= 64 && H % 32 == 0, "add_rmsnorm_img: M=%d (<= 64) H=%d (%% 32 == 0), y_img required", M, H);
    return add_rmsnorm_launch(x_f16, partials, nsplit, ld, bias, residual_in, residual_out, weight, eps, M, H, y_img, cdiv(M, 16), act_dtype, stream, touch);
}

// ---------------------------------------------------------------- activation image
namespace {
// BF: the row-major side is bf16; the image side is always fp16 (common.h)
template <bool BF>
__global__ __launch_bounds__(256) void act_image_pack_kernel(const u32x4* __restrict__ src, int M, int K, u32x4* __restrict__ dst, int mblk, int unpack) {
    const int nvec = K >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * nvec) return;
    const int m = idx / nvec, c0 = (idx - m * nvec) * 8;
    const size_t a = ((size_t)m * K + c0) >> 3, b = act_img_index(m, c0, mblk) >> 3;
    float v[8];
    if (unpack) {
        act_unpack8<false>(src[b], v);
        if constexpr (BF) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= kImgBfUnscale;      // the image of a bf16 tensor holds x 2^-8 (common.h)
        }
        dst[a] = act_pack8<BF>(v);
    } else {
        act_unpack8<BF>(src[a], v);
        dst[b] = img_pack8<BF>(v);
    }
}
} // namespace

extern "C" size_t mi355_act_image_bytes(int32_t M, int32_t K) {
    return (M <= 0 || K <= 0) ? 0 : (size_t)cdiv(M, 16) * 16 * (size_t)((K + 31) & ~31) * 2;
}

// direction 0: row-major x [M][K] of act_dtype -> image (fp16; the image of a bf16 tensor holds x 2^-8, common.h img_val);
// 1: image (of a tensor of act_dtype) -> row-major tensor of act_dtype
extern "C" int mi355_act_image_pack(const void* src, int32_t M, int32_t K, void* dst, int32_t direction, int32_t act_dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(src && dst && M > 0 && M <= 64 && K > 0 && K % 32 == 0, "act_image_pack: M=%d (1..64) K=%d (%% 32 == 0)", M, K);
    MI355_CHECK_ARG(act_dtype == MI355_ACT_F16 || act_dtype == MI355_ACT_BF16, "act_image_pack: act_dtype=%d", act_dtype);
    const dim3 grid(cdiv(M * (K / 8), 256));
    if (act_dtype == MI355_ACT_BF16)
        hipLaunchKernelGGL(act_image_pack_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, M, K, (u32x4*)dst, cdiv(M, 16), direction);
    else
        hipLaunchKernelGGL(act_image_pack_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, M, K, (u32x4*)dst, cdiv(M, 16), direction);
    MI355_CHECK_LAUNCH("act_image_pack_kernel");
    return MI355_OK;
}

extern "C" int mi355_add_rmsnorm(const void* x_f16, const float* partials, int32_t nsplit, int32_t ld, const void* bias,
                                 const void* residual_in, void* residual_out, const void* weight, float eps, int32_t M,
                                 int32_t H, void* y, mi355_stream_t stream) {
    return mi355_add_rmsnorm_dt(x_f16, partials, nsplit, ld, bias, residual_in, residual_out, weight, eps, M, H, y, MI355_ACT_F16, stream);
}

extern "C" int mi355_rmsnorm_dt(const void* x, const void* weight, float eps, int32_t M, int32_t H, void* y, int32_t act_dtype,
                                mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && y, "rmsnorm: null pointer");
    return mi355_add_rmsnorm_dt(x, nullptr, 0, 0, nullptr, nullptr, nullptr, weight, eps, M, H, y, act_dtype, stream);
}

extern "C" int mi355_rmsnorm(const void* x, const void* weight, float eps, int32_t M, int32_t H, void* y,
                             mi355_stream_t stream) {
    return mi355_rmsnorm_dt(x, weight, eps, M, H, y, MI355_ACT_F16, stream);
}

extern "C" int mi355_silu_mul_dt(const void* gate_up, int32_t M, int32_t I, void* out, int32_t act_dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(gate_up && out && M > 0 && I > 0 && I % 8 == 0, "silu_mul: M=%d I=%d", M, I);
    MI355_CHECK_ARG(act_dtype == MI355_ACT_F16 || act_dtype == MI355_ACT_BF16, "silu_mul: act_dtype=%d", act_dtype);
    const int total = M * (I / 8);
    if (act_dtype == MI355_ACT_BF16)
        hipLaunchKernelGGL(silu_mul_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)gate_up, M, I, (f16*)out);
    else
        hipLaunchKernelGGL(silu_mul_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)gate_up, M, I, (f16*)out);
    MI355_CHECK_LAUNCH("silu_mul_kernel");
    return MI355_OK;
}

extern "C" int mi355_silu_mul(const void* gate_up, int32_t M, int32_t I, void* out, mi355_stream_t stream) {
    return mi355_silu_mul_dt(gate_up, M, I, out, MI355_ACT_F16, stream);
}

extern "C" int mi355_embedding(const int32_t* ids, int32_t T, const void* table, int32_t H, int32_t vocab, void* out,
                               mi355_stream_t stream) {
    MI355_CHECK_ARG(ids && table && out && T > 0 && H > 0 && H % 8 == 0 && vocab > 0, "embedding: T=%d H=%d", T, H);
    const int total = T * (H / 8);
    hipLaunchKernelGGL(embedding_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, ids, T, (const f16*)table,
                       H, vocab, (f16*)out);
    MI355_CHECK_LAUNCH("embedding_kernel");
    return MI355_OK;
}

extern "C" int mi355_argmax_ex(const float* logits, int32_t B, int32_t V, int32_t ld, int32_t* ids, int32_t* positions,
                               void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(logits && ids && workspace && B > 0 && V > 0 && ld >= V && ld % 4 == 0, "argmax: B=%d V=%d ld=%d", B, V, ld);
    const int nparts = 64;
    if (workspace_bytes < (size_t)B * nparts * 8) { mi355_set_error("argmax: workspace too small"); return MI355_ERR_WORKSPACE; }
    float* cv = (float*)workspace; int* ci = (int*)(cv + (size_t)B * nparts);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(argmax_stage1, dim3(B, nparts), dim3(256), 0, st, logits, V, ld, cv, ci);
    hipLaunchKernelGGL(argmax_stage2, dim3(B), dim3(64), 0, st, (const float*)cv, (const int*)ci, nparts, ids, positions);
    MI355_CHECK_LAUNCH("argmax");
    return MI355_OK;
}

extern "C" int mi355_prefetch(const void* ptr, size_t bytes, void* sink, mi355_stream_t stream) {
    if (!ptr || bytes < 16) return MI355_OK;
    const size_t nvec = bytes / 16;
    const int grid = (int)(nvec / 256 < 512 ? (nvec + 255) / 256 : 512);
    hipLaunchKernelGGL(prefetch_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ptr, nvec, (uint32_t*)sink);
    MI355_CHECK_LAUNCH("prefetch_kernel");
    return MI355_OK;
}

// stage 1 only: per-row candidates (value, local index) x 64 into `workspace` (consumed by mi355_allreduce_argmax)
extern "C" int mi355_argmax_candidates(const float* logits, int32_t B, int32_t V, int32_t ld, void* workspace,
                                       size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(logits && workspace && B > 0 && V > 0 && ld >= V && ld % 4 == 0, "argmax: B=%d V=%d ld=%d", B, V, ld);
    const int nparts = 64;
    if (workspace_bytes < (size_t)B * nparts * 8) { mi355_set_error("argmax: workspace too small"); return MI355_ERR_WORKSPACE; }
    float* cv = (float*)workspace; int* ci = (int*)(cv + (size_t)B * nparts);
    hipLaunchKernelGGL(argmax_stage1, dim3(B, nparts), dim3(256), 0, (hipStream_t)stream, logits, V, ld, cv, ci);
    MI355_CHECK_LAUNCH("argmax_stage1");
    return MI355_OK;
}

extern "C" int mi355_argmax(const float* logits, int32_t B, int32_t V, int32_t ld, int32_t* ids, void* workspace,
                            size_t workspace_bytes, mi355_stream_t stream) {
    return mi355_argmax_ex(logits, B, V, ld, ids, nullptr, workspace, workspace_bytes, stream);
}

// vocab-split greedy, local half: one (max, global index) pair per row -> pairs_out [B] (8 bytes each)
extern "C" int mi355_argmax_pairs(const float* logits, int32_t B, int32_t V, int32_t ld, int32_t vocab_offset, void* pairs_out,
                                  void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(pairs_out && vocab_offset >= 0, "argmax_pairs: bad argument");
    if (int e = mi355_argmax_candidates(logits, B, V, ld, workspace, workspace_bytes, stream)) return e;
    const float* cv = (const float*)workspace; const int* ci = (const int*)(cv + (size_t)B * 64);
    hipLaunchKernelGGL(argmax_pair, dim3(B), dim3(64), 0, (hipStream_t)stream, cv, ci, 64, vocab_offset, (ArgPair*)pairs_out);
    MI355_CHECK_LAUNCH("argmax_pair");
    return MI355_OK;
}

// global half: pairs_all [world][B] (all-gathered) -> ids [B], positions[b] += 1
extern "C" int mi355_argmax_pick(const void* pairs_all, int32_t world, int32_t B, int32_t* ids, int32_t* positions,
                                 mi355_stream_t stream) {
    MI355_CHECK_ARG(pairs_all && ids && world > 0 && B > 0, "argmax_pick: world=%d B=%d", world, B);
    hipLaunchKernelGGL(argmax_pick, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, (const ArgPair*)pairs_all, world, B, ids,
                       positions);
    MI355_CHECK_LAUNCH("argmax_pick");
    return MI355_OK;
}
