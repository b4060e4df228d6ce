// Entry points shared between translation units of libmi355_decode.so but not part
// of the public C-ABI (include/mi355_decode.h).
#pragma once
#include "../../include/mi355_decode.h"
#ifdef __cplusplus
extern "C" {
#endif
int mi355_gemm_plan(int M, const mi355_weight_t* w, int max_splits, int* nbw_out, int* cps_out);
int mi355_linear_direct(const void* x, int32_t M, const mi355_weight_t* w, const void* bias, void* y, int32_t epilogue,
                        void* workspace, size_t workspace_bytes, mi355_stream_t stream);
int mi355_argmax_ex(const float* logits, int32_t B, int32_t V, int32_t ld, int32_t* ids, int32_t* positions, void* workspace,
                    size_t workspace_bytes, mi355_stream_t stream);
/* seq_lens_minus_one != 0: seq_lens[] holds tokens already cached (decode "positions"), context = value + 1 */
int mi355_paged_decode_attn_ex(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table,
                               int32_t max_blocks_per_seq, const int32_t* seq_lens, int32_t seq_lens_minus_one, int32_t B,
                               int32_t nh, float scale, int32_t max_seq_len, void* out, void* workspace,
                               size_t workspace_bytes, mi355_stream_t stream);
int mi355_fullk_weight_ok(const mi355_weight_t* w);
int mi355_fullk64_weight_ok(const mi355_weight_t* w);     /* gemm.hip: the full-K launches on activation images (gemm_fullk64.hip) take this linear */
int mi355_fullk64_qkv_ok(const mi355_weight_t* w, int32_t hd);   /* ... as the QKV + RoPE launch (also K up to 9600 when its tile pairs leave half the chip free) */
int mi355_gemm_wide_direct_ok(const mi355_weight_t* w);   /* gemm_wide.hip: the one-launch form takes this linear (N fills the chip) */
int mi355_gemm_splitk64_plan(int M, int NT, int KC, int wbits, int group_size, int max_splits, int* cps_out);   /* gemm_splitk64.hip: slabs, or < 0 */
int mi355_gemm_splitk64_direct_plan(int M, int NT, int KC, int wbits, int group_size);   /* gemm_splitk64.hip, direct form: tiles per block, or < 0 */
int mi355_prefetch(const void* ptr, size_t bytes, void* sink, mi355_stream_t stream);
/* Touch plan: the blocks a latency-bound launch has to spare (the slab fold in front of a QKV launch: 64 blocks on 256 CUs) read one dword per
 * 128-byte line of the weights the NEXT launch streams, so that its first requests are served by the Infinity Cache (or the XCD's L2)
 * instead of HBM under the launch burst.  Unit u = what block u of the next launch reads, touched by a block on the same XCD (block b runs
 * on XCD b % 8 -- observed, speed only).  For gemm_fullk64's QKV launch unit u is the (d, d + hd/2) tile pair u: two runs of run_bytes at
 * tiles (u / hh) 2 hh + u % hh and + hh, and the (zero, scale) words of those tiles.  No effect on any result. */
typedef struct {
    const void* qw; const void* meta;     /* weight image, (zero, scale) words [groups][N_pad] */
    uint32_t run_bytes;                   /* bytes of one tile's weights (contiguous) */
    uint32_t meta_groups, meta_stride;    /* groups along K, dwords from group to group */
    int32_t  n_units, hh;                 /* tile pairs, tiles per half head */
    void*    sink;                        /* a dword nobody reads */
    int32_t  delay;                       /* the spare blocks wait delay x 512 cycles first (the launch's own requests go out ahead) */
} mi355_touch_t;
int mi355_qkv_touch_plan(const mi355_weight_t* wqkv, int32_t hd, void* sink, mi355_touch_t* out);   /* gemm.hip: MI355_OK, or MI355_ERR_UNSUPPORTED */
int mi355_add_rmsnorm_img_touch(const void* x_f16, const float* partials, int32_t nsplit, int32_t ld, const void* bias, const void* residual_in,
                                void* residual_out, const void* weight, float eps, int32_t M, int32_t H, void* y_img, int32_t act_dtype,
                                const mi355_touch_t* touch, mi355_stream_t stream);
/* Round 6, tensor parallelism: a row-parallel shard (O / down) written by its full-K GEMM straight into the rank's registered all-reduce buffer.
 * allreduce.hip fills the target (slot layout of the NEXT <= 64-row call of the context), gemm.hip's mi355_linear_publish_img passes it to gemm_fullk64.hip. */
typedef struct {
    const uint32_t* epoch;                /* device: per-block call counters of the context */
    void*    data;                        /* this rank's registered buffer */
    uint32_t bytes, parity_elems, slot_elems;
    int32_t  plain_stores;                /* full-fence hand-over: plain stores (the all-reduce launch's release fence publishes them) */
} mi355_publish_target_t;
int mi355_allreduce_publish_target(mi355_allreduce_t* ar, int32_t T, int32_t H, mi355_publish_target_t* out);
int mi355_fullk64_publish_ok(const mi355_weight_t* w);    /* gemm.hip: mi355_linear_publish_img takes this linear (W4 g128 / per-channel W8, K <= 5760; up to 9600 when its tile pairs leave half the chip free) */
int mi355_argmax_candidates(const float* logits, int32_t B, int32_t V, int32_t ld, void* workspace, size_t workspace_bytes,
                            mi355_stream_t stream);
int mi355_argmax_pairs(const float* logits, int32_t B, int32_t V, int32_t ld, int32_t vocab_offset, void* pairs_out,
                       void* workspace, size_t workspace_bytes, mi355_stream_t stream);
int mi355_argmax_pick(const void* pairs_all, int32_t world, int32_t B, int32_t* ids, int32_t* positions, mi355_stream_t stream);
#ifdef __cplusplus
}
#endif
