// Speculative-decoding verify step: row softmax and chain rejection sampling, gfx950.
//
// Replaces rejection_sampling_kernel / invokeRejectionSampling
// (rtp_llm/models_py/bindings/rocm/speculative_sampling/sampling.cu:306-530, called from
// cpp/normal_engine/speculative/SpeculativeSampler.cc:214 and MtpExecutor.cc:1405).  Argument meaning, output
// padding (-1) and the accepted-count convention (accepted drafts + 1) follow the reference; the residual draw is
// "first index whose inclusive prefix sum of relu(q - p) exceeds u * sum" with the prefix taken in index order.
// One 1024-thread block per row: the accept chain is serial in one lane (gamma <= 16 dependent reads), the residual
// draw is two passes over the vocabulary row (sum, then locate), each thread owning a contiguous segment so that the
// block-level prefix over threads is the prefix over indices.
#include "common.h"

namespace {

constexpr int kThreads = 1024;

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float t = (threadIdx.x < kThreads / 64) ? s_red[threadIdx.x] : 0.f;
    if (wave == 0) {
        t = wave_sum(t);
        if (lane == 0) s_red[0] = t;
    }
    __syncthreads();
    const float r = s_red[0];
    __syncthreads();
    return r;
}

// Inverse-CDF draw from the unnormalised non-negative weights f(0..V-1): the first index j with f(j) > 0 whose inclusive
// prefix sum, taken in index order, exceeds u01 * sum (V - 1 when rounding leaves none: sampling.cu:418).  Thread tx owns
// the contiguous segment [lo, hi), so the block-level prefix over threads is the prefix over indices.  Result in *s_found
// (valid for every thread after the call).
template <class F>
__device__ __forceinline__ void block_draw(F&& f, int V, float u01, float* s_red, float* s_scan, int* s_found) {
    const int tx = threadIdx.x;
    const int seg = (V + kThreads - 1) / kThreads;
    const int lo = min(V, tx * seg), hi = min(V, lo + seg);
    float local = 0.f;
    for (int j = lo; j < hi; ++j) local += f(j);
    const float total = block_sum(local, s_red);
    const float u = u01 * total;
    // inclusive prefix over threads (Hillis-Steele in LDS; 1024 entries)
    s_scan[tx] = local;
    if (tx == 0) *s_found = V - 1;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        const float add = (tx >= off) ? s_scan[tx - off] : 0.f;
        __syncthreads();
        s_scan[tx] += add;
        __syncthreads();
    }
    const float incl = s_scan[tx], excl = incl - local;
    if (local > 0.f && excl <= u && incl > u) {   // the prefix crosses u inside this segment (at most one thread)
        float c = excl;
        int found = hi - 1;
        for (int j = lo; j < hi; ++j) {
            const float r = f(j);
            c += r;
            if (r > 0.f && c > u) { found = j; break; }
        }
        *s_found = found;
    }
    __syncthreads();
}

struct RejectParams {
    const float*   draft_probs;
    const int32_t* draft_ids;
    const float*   uniform;
    const float*   target_probs;
    const int32_t* target_ids;
    int            target_stride;
    int32_t*       out_ids;
    int32_t*       out_accepted;
    const uint8_t* do_sample;
    int            B, gamma, V, point_mass;
};

__global__ __launch_bounds__(kThreads) void rejection_sample_kernel(const RejectParams p) {
    const int row = blockIdx.x, tx = threadIdx.x;
    const int G = p.gamma, G1 = p.gamma + 1;
    __shared__ int   s_pos, s_skip;
    __shared__ float s_red[kThreads / 64];
    __shared__ float s_scan[kThreads];
    __shared__ int   s_found;

    if (tx == 0) { // accept chain (sampling.cu:336-379)
        const bool sample = p.do_sample[row] != 0;
        bool fallback = false;
        int pos = G;
        for (int i = 0; i < G; ++i) {
            const int d = p.draft_ids[row * G + i];
            const int t = p.target_ids[(size_t)(row * G1 + i) * p.target_stride + p.target_stride - 1];
            const float q = p.target_probs[(size_t)(row * G1 + i) * p.V + d];
            const float pr = p.point_mass ? 1.0f : p.draft_probs[(size_t)(row * G + i) * p.V + d];
            const float u = p.uniform[row * G1 + i];
            const bool accept = sample ? (u * pr < q) : (t == d);
            if (accept) {
                p.out_ids[row * G1 + i] = d;
            } else {
                pos = i;
                if (!sample) {
                    p.out_ids[row * G1 + i] = t;
                    for (int n = i + 1; n < G1; ++n) p.out_ids[row * G1 + n] = -1;
                    fallback = true;
                }
                break;
            }
        }
        p.out_accepted[row] = pos + 1;
        if (pos == G) p.out_ids[row * G1 + pos] = p.target_ids[(size_t)(row * G1 + pos) * p.target_stride + p.target_stride - 1];
        s_pos = pos;
        s_skip = (fallback || pos == G) ? 1 : 0;
    }
    __syncthreads();
    if (s_skip) return;
    const int pos = s_pos;

    // residual distribution relu(q - p) of position pos; thread tx owns indices [lo, hi)
    const float* q = p.target_probs + (size_t)(row * G1 + pos) * p.V;
    const float* dp = p.point_mass ? nullptr : p.draft_probs + (size_t)(row * G + pos) * p.V;
    const int dtok = p.draft_ids[row * G + pos];
    auto resid = [&](int j) {
        const float pv = dp ? dp[j] : (j == dtok ? 1.0f : 0.0f);
        return fmaxf(q[j] - pv, 0.f);
    };
    block_draw(resid, p.V, p.uniform[row * G1 + min(pos + 1, G)], s_red, s_scan, &s_found);
    if (tx == 0) {
        p.out_ids[row * G1 + pos] = s_found;
        for (int n = pos + 1; n < G1; ++n) p.out_ids[row * G1 + n] = -1;
    }
}

// probs[r, :] = softmax(logits[r, :] / temperature), fp32 (the sampler's distribution; SpeculativeSampler.cc feeds
// rejection sampling with these rows)
__global__ __launch_bounds__(kThreads) void softmax_rows_kernel(const float* __restrict__ logits, int V, int ld, float inv_temp,
                                                                float* __restrict__ probs) {
    __shared__ float s_red[kThreads / 64];
    const float* x = logits + (size_t)blockIdx.x * ld;
    float* y = probs + (size_t)blockIdx.x * V;
    float m = -3.0e38f;
    for (int j = threadIdx.x; j < V; j += kThreads) m = fmaxf(m, x[j]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_red[0];
    for (int w = 1; w < kThreads / 64; ++w) m = fmaxf(m, s_red[w]);
    __syncthreads();
    float s = 0.f;
    for (int j = threadIdx.x; j < V; j += kThreads) s += __expf((x[j] - m) * inv_temp);
    s = block_sum(s, s_red);
    const float inv = 1.f / s;
    for (int j = threadIdx.x; j < V; j += kThreads) y[j] = __expf((x[j] - m) * inv_temp) * inv;
}

// ids[r] = inverse-CDF sample of probs[r, :] with uniform[r] (the top_k = 0 / top_p = 1 branch of the sampler:
// softmax, then sampling from the probabilities, bindings/core/CudaSampleOp.cc:702-737)
__global__ __launch_bounds__(kThreads) void sample_rows_kernel(const float* __restrict__ probs, int V, int ld,
                                                               const float* __restrict__ uniform, int32_t* __restrict__ ids) {
    __shared__ float s_red[kThreads / 64];
    __shared__ float s_scan[kThreads];
    __shared__ int   s_found;
    const float* q = probs + (size_t)blockIdx.x * ld;
    block_draw([&](int j) { return fmaxf(q[j], 0.f); }, V, uniform[blockIdx.x], s_red, s_scan, &s_found);
    if (threadIdx.x == 0) ids[blockIdx.x] = s_found;
}

} // namespace

extern "C" int mi355_sample_rows(const float* probs, int32_t rows, int32_t V, int32_t ld, const float* uniform_samples,
                                 int32_t* ids, mi355_stream_t stream) {
    MI355_CHECK_ARG(probs && uniform_samples && ids && rows > 0 && V > 0 && ld >= V, "sample_rows: bad args");
    hipLaunchKernelGGL(sample_rows_kernel, dim3(rows), dim3(kThreads), 0, (hipStream_t)stream, probs, V, ld, uniform_samples, ids);
    MI355_CHECK_LAUNCH("sample_rows_kernel");
    return MI355_OK;
}

extern "C" int mi355_softmax_rows(const float* logits, int32_t rows, int32_t V, int32_t ld, float temperature, float* probs,
                                  mi355_stream_t stream) {
    MI355_CHECK_ARG(logits && probs && rows > 0 && V > 0 && ld >= V && temperature > 0.f, "softmax_rows: bad args");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(kThreads), 0, (hipStream_t)stream, logits, V, ld, 1.f / temperature, probs);
    MI355_CHECK_LAUNCH("softmax_rows_kernel");
    return MI355_OK;
}

extern "C" int mi355_rejection_sample(const float* draft_probs, const int32_t* draft_token_ids, const float* uniform_samples,
                                      const float* target_probs, const int32_t* target_token_ids, int32_t target_token_stride,
                                      int32_t* output_token_ids, int32_t* output_accepted_token_num, const uint8_t* do_sample,
                                      int32_t batch_size, int32_t num_speculative_tokens, int32_t vocab_size,
                                      int32_t draft_probs_point_mass, mi355_stream_t stream) {
    if (batch_size == 0) return MI355_OK; // as invokeRejectionSampling (sampling.cu:491-493)
    MI355_CHECK_ARG(draft_token_ids && uniform_samples && target_probs && target_token_ids && output_token_ids &&
                        output_accepted_token_num && do_sample,
                    "rejection_sample: null pointer");
    MI355_CHECK_ARG(draft_probs || draft_probs_point_mass, "rejection_sample: draft_probs missing (and not point-mass)");
    MI355_CHECK_ARG(batch_size > 0 && num_speculative_tokens > 0 && vocab_size > 0 && target_token_stride > 0,
                    "rejection_sample: batch=%d gamma=%d vocab=%d stride=%d", batch_size, num_speculative_tokens, vocab_size,
                    target_token_stride);
    RejectParams p{draft_probs, draft_token_ids, uniform_samples, target_probs, target_token_ids, target_token_stride,
                   output_token_ids, output_accepted_token_num, do_sample, batch_size, num_speculative_tokens, vocab_size,
                   draft_probs_point_mass ? 1 : 0};
    hipLaunchKernelGGL(rejection_sample_kernel, dim3(batch_size), dim3(kThreads), 0, (hipStream_t)stream, p);
    MI355_CHECK_LAUNCH("rejection_sample_kernel");
    return MI355_OK;
}
