// Sampler kernels on fp32 logit rows, gfx950: the speculative-decoding verify step (row softmax, chain rejection sampling) and
// the non-greedy branch of the sampler (temperature + repetition / presence / frequency penalties, top-k / top-p filter,
// inverse-CDF draw: bindings/core/CudaSampleOp.cc:632-786, common/kernels/sampling_penalty_kernels.cu:26-180).
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
// the contiguous segment [lo, hi) (a multiple of 4 long: segments start on 16-byte boundaries of an aligned row), so the block-level
// prefix over threads is the prefix over indices.  Result in *s_found (valid for every thread after the call); returns sum.
__device__ __forceinline__ int draw_segment(int V) { return (((V + kThreads - 1) / kThreads) + 3) & ~3; }

template <class F>
__device__ __forceinline__ float block_draw(F&& f, int V, float u01, float* s_red, float* s_scan, int* s_found) {
    const int tx = threadIdx.x;
    const int seg = draw_segment(V);
    const int lo = min(V, tx * seg), hi = min(V, lo + seg);
    float local = 0.f;
#pragma unroll 4
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
        int found = -1, last_pos = -1;              // the rescan rounds differently from the scanned prefix: if c never exceeds u here,
        for (int j = lo; j < hi; ++j) {             // fall back to the LAST index of the segment with weight > 0 -- never to a token the
            const float r = f(j);                   // filter / ban / residual removed (weight 0)
            c += r;
            if (r > 0.f) { last_pos = j; if (c > u) { found = j; break; } }
        }
        *s_found = found >= 0 ? found : last_pos;   // local > 0: the segment holds at least one positive weight
    }
    __syncthreads();
    return total;
}

// One pass over a row of V floats by the whole block: g(value) for every element, 16-byte loads with four in flight per thread
// when the row allows (a scalar loop waits a full L2 round trip per element: 60 us per pass over a 152064-wide row instead of 4).
template <class G>
__device__ __forceinline__ void row_pass(const float* __restrict__ q, int V, bool vec, G&& g) {
    const int tx = threadIdx.x;
    int done = 0;
    if (vec) {
        const f32x4* q4 = reinterpret_cast<const f32x4*>(q);
        const int V4 = V >> 2;
#pragma unroll 4
        for (int j = tx; j < V4; j += kThreads) {
            const f32x4 v = q4[j];
            g(v[0]); g(v[1]); g(v[2]); g(v[3]);
        }
        done = V4 << 2;
    }
    for (int j = done + tx; j < V; j += kThreads) g(q[j]);
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
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
    float m = -3.0e38f;
    row_pass(x, V, vec, [&](float v) { m = fmaxf(m, v); });
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_red[0];
    for (int w = 1; w < kThreads / 64; ++w) m = fmaxf(m, s_red[w]);
    __syncthreads();
    float s = 0.f;
    row_pass(x, V, vec, [&](float v) { s += __expf((v - m) * inv_temp); });
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


// (a, b) summed over the block in a fixed order (wave butterfly, then 16 partials added by every thread): deterministic.
__device__ __forceinline__ void block_sum2(float& a, float& b, float (*s_pair)[2]) {
    a = wave_sum(a); b = wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_pair[wave][0] = a; s_pair[wave][1] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) { ta += s_pair[w][0]; tb += s_pair[w][1]; }
    __syncthreads();
    a = ta; b = tb;
}

// Temperature, then repetition / presence / frequency penalties, in place on fp32 logits: one block per row.
//   batchApplyTemperaturePenalty (sampling_penalty_kernels.cu:26-54): logit *= 1 / (T[row] + 1e-6)
//   batchApplyPenaltyLongSeq (:129-180): every vocabulary id seen among output_ids[0 .. step)[row] (padding positions
//   [input_length, max_input_length) skipped, ids outside [0, V) skipped) is penalised ONCE:
//   logit = logit < 0 ? logit * rep : logit / rep;  logit -= presence;  logit -= frequency * count.
struct PenaltyParams {
    float*         logits;
    int            V, ld;
    const float   *temperature, *repetition, *presence, *frequency;
    const int32_t* output_ids;     // [step][batch]
    const int32_t* input_lengths;  // [batch] or null
    int            batch, max_input_length, step;
    int32_t*       counts;         // [batch][V], zero on entry
};

__global__ __launch_bounds__(kThreads) void penalties_kernel(const PenaltyParams p) {
    const int row = blockIdx.x, tx = threadIdx.x;
    float* x = p.logits + (size_t)row * p.ld;
    const bool pen = p.output_ids != nullptr;
    int32_t* cnt = pen ? p.counts + (size_t)row * p.V : nullptr;
    // ---- temperature: the whole row, 16-byte accesses
    if (p.temperature) {
        const float inv_t = 1.0f / (p.temperature[row] + 1e-6f);
        int done = 0;
        if ((p.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p.logits) & 15) == 0) {
            f32x4* x4 = reinterpret_cast<f32x4*>(x);
            const int V4 = p.V >> 2;
#pragma unroll 4
            for (int j = tx; j < V4; j += kThreads) {
                f32x4 v = x4[j];
                v[0] *= inv_t; v[1] *= inv_t; v[2] *= inv_t; v[3] *= inv_t;
                x4[j] = v;
            }
            done = V4 << 2;
        }
        for (int j = done + tx; j < p.V; j += kThreads) x[j] *= inv_t;
    }
    if (!pen) return;
    // ---- penalties: count the ids of the history; the lane whose increment found a zero owns the id and penalises it once the
    // counts are final -- the work is proportional to the history, not to the vocabulary (histories beyond 32 entries per lane
    // fall back to a scan of the row)
    const float rep = p.repetition ? p.repetition[row] : 1.0f, pres = p.presence ? p.presence[row] : 0.0f,
                freq = p.frequency ? p.frequency[row] : 0.0f;
    auto penalise = [&](int j, int c) {
        // written by other lanes of this block (temperature pass above): read past the CU's vector cache
        float logit = __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(x + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (p.repetition) logit = logit < 0.0f ? logit * rep : logit / rep;
        if (p.presence) logit -= pres;
        if (p.frequency) logit -= freq * (float)c;
        x[j] = logit;
    };
    const int input_length = p.input_lengths ? p.input_lengths[row] : p.max_input_length;
    uint32_t mine = 0;                                     // bit s: the s-th history entry of this lane was the first of its id
    int s = 0;
    for (int index = tx; index < p.step; index += kThreads, ++s) {
        if (index >= input_length && index < p.max_input_length) continue;
        const int tok = p.output_ids[(size_t)index * p.batch + row];
        if (tok >= p.V || tok < 0) continue;
        const int old = atomicAdd(&cnt[tok], 1);
        if (old == 0 && s < 32) mine |= 1u << s;
    }
    __syncthreads();
    if (p.step <= 32 * kThreads) {
        s = 0;
        for (int index = tx; index < p.step; index += kThreads, ++s) {
            if (!((mine >> s) & 1u)) continue;
            const int tok = p.output_ids[(size_t)index * p.batch + row];
            penalise(tok, __hip_atomic_load(&cnt[tok], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    } else {
        for (int j = tx; j < p.V; j += kThreads) {
            const int c = __hip_atomic_load(&cnt[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c > 0) penalise(j, c);
        }
    }
}

// Top-k / top-p filter and draw on probability rows (CudaSampleOp.cc:748-786), one block per row:
//   k = top_k[row] (<= 0 or >= V: no filter): entries below the k-th largest value are dropped (ties with it stay, :755-760);
//   p = top_p[row] (|p| < 1e-7 reads as 1; |p - 1| < 1e-7: no filter): in descending order (equal values in index order) an
//       entry is dropped when the mass before it exceeds p (:768-775).  Only the smallest surviving value v can lose part of
//       its equals: with S = mass of all strictly larger entries, its j-th occurrence stays while S + j v <= p;
//   the survivors are renormalised by max(sum, 1e-10) (probs_out, optional) and one index is drawn by inverse CDF in index order
//   with uniform[row] (torch.multinomial in the reference: same distribution, explicit randomness).
// No sort: both thresholds are built bit by bit from the top of the fp32 pattern (non-negative floats order like their bits), one
// pass over the row and one fixed-order block reduction per bit.
struct TopKPParams {
    const float*   probs;
    int            V, ld;
    const int32_t* top_k;
    const float*   top_p;
    const float*   uniform;
    int32_t*       ids;
    float*         probs_out;
    int            ld_out;
};

__global__ __launch_bounds__(kThreads) void top_k_top_p_sample_kernel(const TopKPParams p) {
    __shared__ float s_red[kThreads / 64];
    __shared__ float s_pair[kThreads / 64][2];
    __shared__ float s_scan[kThreads];
    __shared__ int   s_found;
    const int row = blockIdx.x, tx = threadIdx.x, V = p.V;
    const float* q = p.probs + (size_t)row * p.ld;
    auto pos = [](float v) { return v > 0.f ? v : 0.f; };   // +0 for -0, negatives and NaN: the bit pattern is the order
    int k = p.top_k ? p.top_k[row] : 0;
    const bool use_k = k > 0 && k < V;
    float tp = p.top_p ? p.top_p[row] : 1.0f;
    if (fabsf(tp) < 1e-7f) tp = 1.0f;
    bool use_p = fabsf(tp - 1.0f) >= 1e-7f;
    uint32_t tk = 0, tq = 0;   // tk: largest t with #(bits >= t) >= k;  tq: largest t with mass(bits > t) > p
    const bool vec = ((p.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.probs) & 15) == 0);
    if (use_p) {               // nothing to drop when even the mass above 0 fits
        float m = 0.f, z = 0.f;
        row_pass(q, V, vec, [&](float x) { m += pos(x); });
        block_sum2(m, z, s_pair);
        use_p = m > tp;
    }
    if (use_k || use_p) {
        for (int bit = 30; bit >= 0; --bit) {
            const uint32_t ck = tk | (1u << bit), cq = tq | (1u << bit);
            float cnt = 0.f, mass = 0.f;
            row_pass(q, V, vec, [&](float x) {
                const float v = pos(x);
                const uint32_t b = __float_as_uint(v);
                cnt += (b >= ck) ? 1.f : 0.f;
                mass += (b > cq) ? v : 0.f;
            });
            block_sum2(cnt, mass, s_pair);
            if (use_k && cnt >= (float)k) tk = ck;
            if (use_p && mass > tp) tq = cq;
        }
    }
    // top-p boundary value vb = tq + 1 (an entry's value: the mass above a threshold only changes at entries).  Its equals are
    // ranked in index order over the contiguous segments block_draw uses; `cut` = first index of this thread's segment from
    // which they are dropped.
    const uint32_t kb = use_k ? tk : 0u, vbits = use_p ? tq + 1u : 0u;
    const int seg = draw_segment(V);
    const int lo = min(V, tx * seg), hi = min(V, lo + seg);
    int cut = hi;
    if (use_p) {
        float sb = 0.f, ties = 0.f;
        for (int j = lo; j < hi; ++j) {
            const float v = pos(q[j]);
            const uint32_t b = __float_as_uint(v);
            sb += (b > vbits) ? v : 0.f;
            ties += (b == vbits) ? 1.f : 0.f;
        }
        const float mine = ties;
        float z = 0.f;
        block_sum2(sb, z, s_pair);
        s_scan[tx] = mine;                       // inclusive scan of the tie counts over threads (exact: counts < 2^24)
        __syncthreads();
        for (int off = 1; off < kThreads; off <<= 1) {
            const float add = (tx >= off) ? s_scan[tx - off] : 0.f;
            __syncthreads();
            s_scan[tx] += add;
            __syncthreads();
        }
        float rank = s_scan[tx] - mine;
        __syncthreads();
        const float vb = __uint_as_float(vbits);
        for (int j = lo; j < hi; ++j) {
            if (__float_as_uint(pos(q[j])) != vbits) continue;
            if (sb + rank * vb > tp) { cut = j; break; }
            rank += 1.f;
        }
    }
    auto kept = [&](int j) {      // j inside this thread's segment
        const float v = pos(q[j]);
        const uint32_t b = __float_as_uint(v);
        return (b >= kb && (b > vbits || (b == vbits && j < cut))) ? v : 0.f;
    };
    const float total = block_draw(kept, V, p.uniform[row], s_red, s_scan, &s_found);
    if (tx == 0) p.ids[row] = s_found;
    if (p.probs_out) {
        const float den = fmaxf(total, 1e-10f);
        float* y = p.probs_out + (size_t)row * p.ld_out;
        for (int j = lo; j < hi; ++j) y[j] = kept(j) / den;
    }
}


// no_repeat_ngram_size: ban_repeat_ngram (bindings/common/kernels/banRepeatNgram.cu:30-136, greedy case beam_width = 1; called from
// CudaSampleOp.cc:242-283).  With N = last_index[row] + 1 tokens so far and n = ngram[row]: every position i <= N - n whose n - 1
// tokens equal the last n - 1 tokens of the sequence bans the token that followed it there (logit = -inf), so that no n-gram of
// the sequence can be produced a second time.  n == 0 or N < n: nothing.  One block per row; the history is read from global
// memory (N n int32 reads per row: nothing next to the vocabulary row the sampler touches anyway).
__global__ __launch_bounds__(256) void ban_repeat_ngram_kernel(float* __restrict__ logits, int V, int ld, const int32_t* __restrict__ token_ids,
                                                               int token_ld, const int32_t* __restrict__ last_index,
                                                               const int32_t* __restrict__ ngram) {
    const int row = blockIdx.x;
    const int n = ngram[row], N = last_index[row] + 1;
    if (n <= 0 || N < n || N > token_ld) return;
    const int32_t* tok = token_ids + (size_t)row * token_ld;
    const int32_t* last = tok + (N - n + 1);                // the n - 1 most recent tokens
    for (int i = threadIdx.x; i <= N - n; i += 256) {
        bool match = true;
        for (int k = 0; k < n - 1; ++k)
            if (tok[i + k] != last[k]) { match = false; break; }
        if (!match) continue;
        const int banned = tok[i + n - 1];
        if (banned >= 0 && banned < V) logits[(size_t)row * ld + banned] = -INFINITY;
    }
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

extern "C" int mi355_apply_penalties(float* logits, int32_t batch_size, int32_t V, int32_t ld, const float* temperature,
                                     const float* repetition_penalty, const float* presence_penalty, const float* frequency_penalty,
                                     const int32_t* output_ids, const int32_t* input_lengths, int32_t max_input_length,
                                     int32_t step, int32_t* penalty_ws, mi355_stream_t stream) {
    if (batch_size == 0) return MI355_OK;
    MI355_CHECK_ARG(logits && batch_size > 0 && V > 0 && ld >= V, "apply_penalties: bad args");
    const bool pen = repetition_penalty || presence_penalty || frequency_penalty;
    MI355_CHECK_ARG(!pen || (output_ids && penalty_ws && step >= 0 && max_input_length >= 0),
                    "apply_penalties: penalties need output_ids [step][batch] and a [batch][V] int32 workspace");
    if (!pen && !temperature) return MI355_OK;
    if (pen) {
        if (hipMemsetAsync(penalty_ws, 0, (size_t)batch_size * V * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
            (void)hipGetLastError();
            mi355_set_error("apply_penalties: hipMemsetAsync failed");
            return MI355_ERR_HIP;
        }
    }
    PenaltyParams p{logits, V, ld, temperature, repetition_penalty, presence_penalty, frequency_penalty,
                    pen ? output_ids : nullptr, input_lengths, batch_size, max_input_length, step, penalty_ws};
    hipLaunchKernelGGL(penalties_kernel, dim3(batch_size), dim3(kThreads), 0, (hipStream_t)stream, p);
    MI355_CHECK_LAUNCH("penalties_kernel");
    return MI355_OK;
}

extern "C" int mi355_top_k_top_p_sample(const float* probs, int32_t rows, int32_t V, int32_t ld, const int32_t* top_k,
                                        const float* top_p, const float* uniform_samples, int32_t* ids, float* probs_out,
                                        int32_t ld_out, mi355_stream_t stream) {
    if (rows == 0) return MI355_OK;
    MI355_CHECK_ARG(probs && uniform_samples && ids && rows > 0 && V > 0 && ld >= V, "top_k_top_p_sample: bad args");
    MI355_CHECK_ARG(!probs_out || ld_out >= V, "top_k_top_p_sample: ld_out %d < V %d", ld_out, V);
    TopKPParams p{probs, V, ld, top_k, top_p, uniform_samples, ids, probs_out, ld_out};
    hipLaunchKernelGGL(top_k_top_p_sample_kernel, dim3(rows), dim3(kThreads), 0, (hipStream_t)stream, p);
    MI355_CHECK_LAUNCH("top_k_top_p_sample_kernel");
    return MI355_OK;
}

extern "C" int mi355_ban_repeat_ngram(float* logits, int32_t batch_size, int32_t V, int32_t ld, const int32_t* token_ids,
                                      int32_t token_ld, const int32_t* sequence_last_index, const int32_t* no_repeat_ngram_size,
                                      mi355_stream_t stream) {
    if (batch_size == 0) return MI355_OK;
    MI355_CHECK_ARG(logits && token_ids && sequence_last_index && no_repeat_ngram_size && batch_size > 0 && V > 0 && ld >= V && token_ld > 0,
                    "ban_repeat_ngram: bad args");
    hipLaunchKernelGGL(ban_repeat_ngram_kernel, dim3(batch_size), dim3(256), 0, (hipStream_t)stream, logits, V, ld, token_ids, token_ld,
                       sequence_last_index, no_repeat_ngram_size);
    MI355_CHECK_LAUNCH("ban_repeat_ngram_kernel");
    return MI355_OK;
}
