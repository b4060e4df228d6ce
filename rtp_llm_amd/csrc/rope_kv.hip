// (split-K reduce +) bias + RoPE + Q-extract + paged KV-cache write for decode, gfx950.
//
// Replaces FusedRopeKVCacheDecodeOp::forward (rtp_llm/models_py/bindings/rocm/
// FusedRopeKVCacheOp.cc:519-646 -> add_fusedQKV_bias_transpose_decode_kernel,
// rocm/kernels/fused_rope_kvcache_kernel.cu:1297-1466) and re-instates the INT8
// KV branch the reference removed: per-(token, kv-head) fp32 scale
// = max|x|/127, round-to-nearest-even + saturate (rocm_utils/_cast_to_int8.h:5-24), bytes stored as code + 128,
// scale plane [block][K|V][nkv][page] (kv_cache_utils.h:265-271).
//
// One wave per (token, head): lane i owns the NeoX pair (i, i + hd/2)
// (rotary_position_embedding.h:278-321,444-449: x' = cos*x - sin*y, y' = cos*y + sin*x
// in fp32, one rounding to fp16).  Launch-bound by nature (T*(nh+2nkv) waves).
#include "common.h"

namespace {

struct RopeParams {
    const f16*     qkv;
    const float*   partials;
    int            nsplit, ld;
    const f16*     bias;
    const float*   cos_sin;
    const int32_t* positions;
    const int32_t* block_table;
    int            max_blocks, T, nh, nkv, hd, page, max_pos, num_blocks, q_len;
    int32_t*       oob_count;
    void*          kv_base;
    float*         scale_base;
    int            kv_int8;
    f16*           q_out;
};

// BF: qkv / bias / q_out and a 16-bit cache are bf16 (kv_dtype MI355_KV_BF16); the INT8 cache is quantised from the same fp32 values either way
template <bool BF>
__global__ __launch_bounds__(256) void rope_kv_write_kernel(const RopeParams p) {
    using raw16 = uint16_t;
    const int t    = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int h    = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int nheads = p.nh + 2 * p.nkv;
    if (h >= nheads) return;
    const int half = p.hd >> 1;
    const bool act = lane < half;
    const int col0 = h * p.hd + lane, col1 = col0 + half;

    // Everything is requested before the first wait (round 2 walked 8 dependent round trips: 4 slabs, 3 tail slabs one by
    // one, bias, position, rotation row, block id): the position and the block id are wave-uniform (scalar loads), the
    // slabs of a column are up to 16 independent loads, the rotation row follows the position.
    const int pos_in = p.positions[t];
    const int pos_lim = min(p.max_pos, p.max_blocks * p.page);
    const int pos = min(max(pos_in, 0), pos_lim - 1);       // clamped: the rotation table and the block table stay in range
    const bool is_v = h >= p.nh + p.nkv;
    const int blk = (h >= p.nh) ? p.block_table[(size_t)(t / p.q_len) * p.max_blocks + pos / p.page] : 0;
    float2 cs = {1.f, 0.f};
    if (!is_v && act) cs = *reinterpret_cast<const float2*>(p.cos_sin + ((size_t)pos * half + lane) * 2);
    float x0 = 0.f, x1 = 0.f;
    if (act) {
        float bh0 = 0.f, bh1 = 0.f;
        const raw16* bias = reinterpret_cast<const raw16*>(p.bias);
        if (p.partials) {
            const size_t sstride = (size_t)p.T * p.ld;
            const float* src = p.partials + (size_t)t * p.ld;
            for (int s0 = 0; s0 < p.nsplit; s0 += 8) {      // 8 slabs (16 loads) in flight per round trip, summed in index order
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {               // slabs past the end: re-read the last one, add zero
                    const int su = min(s0 + u, p.nsplit - 1);
                    a[u] = src[su * sstride + col0]; b[u] = src[su * sstride + col1];
                }
                if (s0 == 0 && p.bias) { bh0 = act_from_bits<BF>(bias[col0]); bh1 = act_from_bits<BF>(bias[col1]); }   // behind the slabs in the (in-order) queue
#pragma unroll
                for (int u = 0; u < 8; ++u) { const bool ok = s0 + u < p.nsplit; x0 += ok ? a[u] : 0.f; x1 += ok ? b[u] : 0.f; }
            }
        } else {
            const raw16* qkv = reinterpret_cast<const raw16*>(p.qkv);
            const raw16 q0 = qkv[(size_t)t * p.ld + col0], q1 = qkv[(size_t)t * p.ld + col1];
            if (p.bias) { bh0 = act_from_bits<BF>(bias[col0]); bh1 = act_from_bits<BF>(bias[col1]); }
            x0 = act_from_bits<BF>(q0); x1 = act_from_bits<BF>(q1);
        }
        x0 += bh0; x1 += bh1;
        // the QKV linear's output is a 16-bit tensor in the reference
        x0 = act_round<BF>(x0); x1 = act_round<BF>(x1);
    }
    if (!is_v && act) {
        const float r0 = cs.x * x0 - cs.y * x1;
        const float r1 = cs.x * x1 + cs.y * x0;
        x0 = act_round<BF>(r0); x1 = act_round<BF>(r1);
    }
    if (h < p.nh) {
        if (act) {
            raw16* dst = reinterpret_cast<raw16*>(p.q_out) + ((size_t)t * p.nh + h) * p.hd;
            dst[lane] = act_to_bits<BF>(x0); dst[lane + half] = act_to_bits<BF>(x1);
        }
        return;
    }
    // ---- K / V into the paged cache
    const int kh  = is_v ? h - p.nh - p.nkv : h - p.nh;
    if (pos_in < 0) return;                                  // padding row of a multi-row step: nothing to store
    if (pos != pos_in || blk < 0 || blk >= p.num_blocks) {   // stale position / block id: never write somebody else's page
        if (h == p.nh && lane == 0 && p.oob_count) atomicAdd(p.oob_count, 1);
        return;
    }
    const int tok = pos % p.page;
    const size_t head_elems = (size_t)p.page * p.hd;
    const size_t blk_base   = ((size_t)blk * 2 + (is_v ? 1 : 0)) * p.nkv + kh; // in units of heads
    int s0, s1; // element offsets inside this head's [page*hd] region
    if (!is_v) { s0 = tok * p.hd + lane; s1 = s0 + half; }
    else       { s0 = lane * p.page + tok; s1 = (lane + half) * p.page + tok; }
    if (!p.kv_int8) {
        if (act) {
            raw16* dst = (raw16*)p.kv_base + blk_base * head_elems;
            dst[s0] = act_to_bits<BF>(x0); dst[s1] = act_to_bits<BF>(x1);
        }
    } else {
        float amax = act ? fmaxf(fabsf(x0), fabsf(x1)) : 0.f;
        amax = wave_max(amax);
        const float scale = amax > 0.f ? amax / 127.f : 1.f;
        if (act) {
            int8_t* dst = (int8_t*)p.kv_base + blk_base * head_elems;
            const float q0 = fminf(fmaxf(rintf(x0 / scale), -128.f), 127.f);
            const float q1 = fminf(fmaxf(rintf(x1 / scale), -128.f), 127.f);
            // stored offset-binary (code + 128): the attention kernel widens bytes under an fp16 exponent, which wants
            // unsigned bytes -- flipping the top bit here (2 lanes x 1 op per token) saves it 2 VALU per 8 bytes there
            dst[s0] = (int8_t)((int)q0 ^ 0x80); dst[s1] = (int8_t)((int)q1 ^ 0x80);
        }
        if (lane == 0) p.scale_base[blk_base * p.page + tok] = scale;
    }
}

} // namespace

extern "C" int mi355_rope_kv_write(const void* qkv_f16, const float* partials, int32_t nsplit, int32_t ld,
                                   const void* qkv_bias, const float* cos_sin, int32_t rope_dim, int32_t max_pos,
                                   const int32_t* positions, const int32_t* block_table, int32_t max_blocks_per_seq,
                                   int32_t T, int32_t nh, const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count,
                                   mi355_stream_t stream) {
    return mi355_rope_kv_write_rows(qkv_f16, partials, nsplit, ld, qkv_bias, cos_sin, rope_dim, max_pos, positions, block_table,
                                    max_blocks_per_seq, T, 1, nh, kv, q_out, oob_count, stream);
}

extern "C" int mi355_rope_kv_write_rows(const void* qkv_f16, const float* partials, int32_t nsplit, int32_t ld,
                                        const void* qkv_bias, const float* cos_sin, int32_t rope_dim, int32_t max_pos,
                                        const int32_t* positions, const int32_t* block_table, int32_t max_blocks_per_seq,
                                        int32_t T, int32_t q_len, int32_t nh, const mi355_kv_layer_t* kv, void* q_out,
                                        int32_t* oob_count, mi355_stream_t stream) {
    MI355_CHECK_ARG(q_len >= 1 && T % q_len == 0, "rope_kv_write: T=%d must be a multiple of q_len=%d", T, q_len);
    MI355_CHECK_ARG((qkv_f16 != nullptr) != (partials != nullptr), "rope_kv_write: exactly one of qkv_f16 / partials");
    MI355_CHECK_ARG(kv && kv->kv_base && cos_sin && positions && block_table && q_out, "rope_kv_write: null pointer");
    MI355_CHECK_ARG(kv->hd == 64 || kv->hd == 128, "rope_kv_write: hd=%d (64 or 128)", kv->hd);
    MI355_CHECK_ARG(rope_dim == kv->hd, "rope_kv_write: rope_dim=%d must equal hd=%d", rope_dim, kv->hd);
    MI355_CHECK_ARG(kv->page > 0 && T > 0 && nh > 0 && kv->nkv > 0 && max_pos > 0 && max_blocks_per_seq > 0 && kv->num_blocks > 0,
                    "rope_kv_write: bad dims");
    MI355_CHECK_ARG(kv->kv_dtype == MI355_KV_FP16 || kv->kv_dtype == MI355_KV_BF16 || (kv->kv_dtype == MI355_KV_INT8 && kv->scale_base),
                    "rope_kv_write: int8 cache needs scale_base");
    const bool bf = kv->act_dtype == MI355_ACT_BF16;
    MI355_CHECK_ARG((kv->act_dtype == MI355_ACT_F16 || bf) && (kv->kv_dtype == MI355_KV_INT8 || bf == (kv->kv_dtype == MI355_KV_BF16)),
                    "rope_kv_write: act_dtype=%d with kv_dtype=%d (a 16-bit cache has the dtype of the activations)", kv->act_dtype, kv->kv_dtype);
    const int nheads = nh + 2 * kv->nkv;
    MI355_CHECK_ARG(ld >= nheads * kv->hd, "rope_kv_write: ld=%d", ld);
    RopeParams p;
    p.qkv = (const f16*)qkv_f16; p.partials = partials; p.nsplit = nsplit; p.ld = ld; p.bias = (const f16*)qkv_bias;
    p.cos_sin = cos_sin; p.positions = positions; p.block_table = block_table; p.max_blocks = max_blocks_per_seq;
    p.T = T; p.nh = nh; p.nkv = kv->nkv; p.hd = kv->hd; p.page = kv->page; p.kv_base = kv->kv_base;
    p.max_pos = max_pos; p.num_blocks = kv->num_blocks; p.oob_count = oob_count; p.q_len = q_len;
    p.scale_base = kv->scale_base; p.kv_int8 = kv->kv_dtype == MI355_KV_INT8; p.q_out = (f16*)q_out;
    if (bf)
        hipLaunchKernelGGL(rope_kv_write_kernel<true>, dim3(T, cdiv(nheads, 4)), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(rope_kv_write_kernel<false>, dim3(T, cdiv(nheads, 4)), dim3(256), 0, (hipStream_t)stream, p);
    MI355_CHECK_LAUNCH("rope_kv_write_kernel");
    return MI355_OK;
}
