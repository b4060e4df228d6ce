/*
 * mi355_decode.h — C-ABI of libmi355_decode.so
 *
 * MI355X (gfx950 / CDNA4) native implementation of the quantized transformer
 * decode hot path of alibaba/rtp-llm: weight-only INT4/INT8 dequant-GEMM,
 * paged flash-decoding attention with fp16 / INT8 KV-cache, fused
 * RMSNorm / RoPE+KV-write / SiLU-gate epilogues, greedy sampling, and a
 * C++ decode-step driver that replays the whole step as one hipGraph.
 *
 * Conventions (mirrors the one C-ABI precedent in the reference,
 * rtp_llm/models_py/bindings/rocm/kernels/pa_decode_dot_kernel.h:9 —
 * raw device pointers, explicit stream, integer status):
 *   - every pointer marked "dev" is a HIP device pointer; 16-bit float
 *     tensors are IEEE fp16 unless the call carries MI355_ACT_BF16 / MI355_KV_BF16
 *     (see "activation dtype" below); tensors are dense row-major unless stated;
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued,
 *     nothing synchronises, nothing allocates (graph-capturable, the
 *     constraint the reference puts on ops under its HIP-graph runner,
 *     rtp_llm/cpp/cuda_graph/cuda_graph_runner.cc:1139-1203);
 *   - return value: 0 = ok, <0 = error (see MI355_ERR_*); no C++ exception
 *     crosses this boundary; mi355_last_error() gives a thread-local message.
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef MI355_DECODE_H
#define MI355_DECODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 3

typedef void* mi355_stream_t; /* hipStream_t */

enum {
    MI355_OK              = 0,
    MI355_ERR_ARG         = -1, /* bad argument / unsupported shape */
    MI355_ERR_HIP         = -2, /* HIP runtime error at launch */
    MI355_ERR_UNSUPPORTED = -3,
    MI355_ERR_WORKSPACE   = -4  /* workspace too small */
};

/* weight element kinds */
enum { MI355_W4 = 4, MI355_W8 = 8, MI355_W16 = 16 };
/* KV-cache element kinds (reference: KvCacheDataType, cpp/model_utils/AttentionConfig.h:9-12;
 * value 1 is the INT8 slot the reference removed, kept here) */
enum { MI355_KV_FP16 = 0, MI355_KV_INT8 = 1, MI355_KV_BF16 = 2 };
/* activation dtype of a call (x / bias / y of a linear, the rows of a norm, Q and the attention output): the reference path is
 * fp16 / bf16 throughout (f16_linear.py:100-112; dtype grid of modules/base/rocm/test/rocm_norm_test.py).  bf16 is carried by
 *   mi355_weight_t.act_dtype (linears), mi355_kv_layer_t.act_dtype (RoPE / KV write / attention: Q, K, V and the output; a 16-bit
 *   cache is then MI355_KV_BF16, the INT8 cache serves both), the *_dt entry points of the norm family and mi355_model_config_t.act_dtype (step driver).
 * bf16 takes W4 group-wise, W8 and 16-bit (then bf16) weights and a bf16 or INT8 KV cache.
 * Tensor parallelism: the all-reduce family has *_dt forms and the RCCL transport a bf16 callback. */
enum { MI355_ACT_F16 = 0, MI355_ACT_BF16 = 1 };

int         mi355_abi_version(void);
const char* mi355_last_error(void);

/* ------------------------------------------------------------------------
 * Packed weight descriptor.
 *
 * Replaces the reference's DenseWeights{kernel,bias,scales,zeros}
 * (rtp_llm/cpp/models/models_weight/Weights.h:23-29) for one linear layer.
 * `qweight` is the MI355-native tile image produced at load time by
 * rtp_llm_amd.quant (the counterpart of RocmImpl.preprocess_groupwise_weight_params,
 * rtp_llm/device/device_impl.py:797-868, and apply_int8, :211-222):
 *
 *   tile (nt, c) covers output columns [16nt, 16nt+16) x input rows
 *   [128c, 128c+128) and is stored at byte offset
 *       ((nt * (K_pad/128) + c) * LPC) * 1024,   LPC = wbits/4 (1, 2, 4)
 *   as LPC wave-loads of 64 lanes x 16 bytes.  Lane l = (i = l & 15, q = l >> 4)
 *   owns column 16nt+i; MFMA k-step s (0..3) of the chunk uses rows
 *       k = 128c + 32s + 8q + e,  e = 0..7.
 *   W4 : dword s of the lane's 16 bytes holds the 8 unsigned nibbles of step s,
 *        nibble e at bit 4*(e/2) + 16*(e&1)   (pairs land in fp16x2 lanes);
 *   W8 : wave-load p (0..1) holds steps 2p and 2p+1, 8 offset-binary bytes each
 *        (stored byte = q + 128);
 *   W16: wave-load s holds the 8 fp16 values of step s.
 *
 * `meta` (W4/W8) is fp16x2 {zneg, scale} per (group, column):
 *       meta[g * N_pad + n],  zneg = -(1024 + z_eff), W = scale * (u - z_eff)
 *   with u the stored unsigned code.  GPTQ: z_eff = z + 1 (GPTQ_FLAG,
 *   device_impl.py:252); AWQ: z_eff = z; symmetric autoquant: z_eff = 8 / 128.
 *   group_size 0 means one group spanning all of K (per-channel scale,
 *   reference a1: device_impl.py:183-192).
 * ---------------------------------------------------------------------- */
typedef struct {
    const void* qweight; /* dev */
    const void* meta;    /* dev, NULL for W16 */
    int32_t     wbits;   /* MI355_W4 / W8 / W16 */
    int32_t     K;       /* logical input features */
    int32_t     N;       /* logical output features */
    int32_t     K_pad;   /* multiple of 128 */
    int32_t     N_pad;   /* multiple of 16 */
    int32_t     group_size; /* 32, 64, 128, or 0 = per-channel */
    int32_t     act_dtype;  /* MI355_ACT_F16 / MI355_ACT_BF16: dtype of x, bias, y (and of the elements of a W16 weight) */
} mi355_weight_t;

/* epilogue flags for mi355_linear_forward */
enum {
    MI355_EPI_NONE     = 0,
    MI355_EPI_SILU_MUL = 1, /* columns are interleaved (gate,up) pairs; y is [M, N/2] */
    MI355_EPI_OUT_F32  = 2, /* y is fp32 [M, N] (lm_head logits) */
    MI355_EPI_OUT_IMAGE = 4, /* mi355_linear_deferred_norm_img / mi355_linear_direct_img: the 16-bit y is written as an activation image (mi355_act_image_*) */
    /* kernel-family hints (per call, for A/B tests; results stay within the same tolerance): */
    MI355_HINT_STAGED        = 0x100, /* 16 < M <= 64: take the LDS-staged kernel instead of the register-resident one */
    MI355_HINT_NO_PERSISTENT = 0x200  /* M <= 8: skip the persistent x-resident kernel */
};

/* Workspace needed by mi355_linear_forward for a given (M, weight). */
size_t mi355_linear_workspace_bytes(int32_t M, const mi355_weight_t* w);

/*
 * y[M,N] = x[M,K] @ W[K,N] (+ bias[N]) — the op behind LinearBase.forward
 * (rtp_llm/models_py/modules/factory/linear/linear_base.py:75-85; only ROCm
 * 16-bit impl: impl/rocm/f16_linear.py:100-112).  fp32 accumulation on MFMA
 * (v_mfma_f32_16x16x32_f16), output rounded once to fp16 (or fp32 with
 * MI355_EPI_OUT_F32, the lm_head contract of PyWrappedModel.cc:1039-1047).
 * M may be any positive value: M <= 64 takes the HBM-bound decode kernels, M >= 128 (prefill chunks) the compute-shaped
 * kernel that reads every weight once per 128 rows, anything between runs as 64-row slabs.
 */
int mi355_linear_forward(const void* x, int32_t M, const mi355_weight_t* w, const void* bias,
                         void* y, int32_t epilogue, void* workspace, size_t workspace_bytes,
                         mi355_stream_t stream);

/*
 * Split-K half of the same GEMM: writes fp32 partial slabs
 * partials[split][M][N_pad] and returns the slab count (>0) — consumed by
 * mi355_rope_kv_write / mi355_add_rmsnorm, which fold the reduction into the
 * next op.  `max_splits` bounds the slab count the caller has room for.
 */
int mi355_linear_partial(const void* x, int32_t M, const mi355_weight_t* w, float* partials,
                         int32_t max_splits, mi355_stream_t stream);

/* ------------------------------------------------------------------------
 * RMSNorm family — replaces RMSNorm / RMSResNorm
 * (rtp_llm/models_py/modules/base/rocm/norm.py:51-77; torch spec
 * modules/base/common/norm.py:83-92: fp32 x*rsqrt(mean(x^2)+eps), cast to
 * fp16, then multiply by weight).
 *
 * x_sum = (x_f16 or sum of nsplit fp32 slabs [nsplit][M][ld]) (+ bias) (+ residual)
 * if residual_out: residual_out = fp16(x_sum)
 * y = weight * fp16(normalise(x_sum))
 * ---------------------------------------------------------------------- */
int mi355_rmsnorm(const void* x, const void* weight, float eps, int32_t M, int32_t H, void* y,
                  mi355_stream_t stream);
/* the same family with the rows / weight / bias in act_dtype (MI355_ACT_F16 = the entry points without the suffix, MI355_ACT_BF16);
 * mi355_embedding copies 16-bit rows and serves both */
int mi355_rmsnorm_dt(const void* x, const void* weight, float eps, int32_t M, int32_t H, void* y, int32_t act_dtype,
                     mi355_stream_t stream);
int mi355_add_rmsnorm_dt(const void* x, const float* partials, int32_t nsplit, int32_t ld, const void* bias,
                         const void* residual_in, void* residual_out, const void* weight, float eps, int32_t M, int32_t H,
                         void* y, int32_t act_dtype, mi355_stream_t stream);
int mi355_silu_mul_dt(const void* gate_up, int32_t M, int32_t I, void* out, int32_t act_dtype, mi355_stream_t stream);

int mi355_add_rmsnorm(const void* x_f16, const float* partials, int32_t nsplit, int32_t ld,
                      const void* bias, const void* residual_in, void* residual_out,
                      const void* weight, float eps, int32_t M, int32_t H, void* y,
                      mi355_stream_t stream);

/* out[M,I] = silu(gate_up[:, :I]) * gate_up[:, I:] — FusedSiluAndMul
 * (modules/base/rocm/activation.py:9-24). */
int mi355_silu_mul(const void* gate_up, int32_t M, int32_t I, void* out, mi355_stream_t stream);

/* out[T,H] = table[ids[T]] — Embedding (modules/base/common/embedding.py:35-49,
 * kernel bindings/common/kernels/embedding_kernels.cu:90). */
int mi355_embedding(const int32_t* ids, int32_t T, const void* table, int32_t H, int32_t vocab,
                    void* out, mi355_stream_t stream);

/* ------------------------------------------------------------------------
 * Paged KV cache, one layer.  Same allocator contract as the reference
 * (per-layer tensor [blocks, 2, nkv, page, hd], K block = pool index 2b,
 * V block = 2b+1: bindings/OpDefs.h:24-28,201-202, kv_cache_kernels.cu:58-61;
 * 8-bit scale plane fp32 [blocks][2][nkv][page]: kv_cache_utils.h:265-271,
 * MHAKVCacheSpec.h:52-54,82-92).  Intra-block layout is native:
 *   K block: [nkv][page][hd]       (token-major, hd contiguous)
 *   V block: [nkv][hd][page]       (channel-major: the MFMA k-dim of P.V is
 *                                   tokens, so tokens are contiguous per channel —
 *                                   same as the reference's NonAsm V layout,
 *                                   kv_cache_utils.h:236-241)
 * ---------------------------------------------------------------------- */
typedef struct {
    void*   kv_base;    /* dev: fp16 or int8 elements */
    float*  scale_base; /* dev: INT8 only, else NULL */
    int32_t kv_dtype;   /* MI355_KV_FP16 / MI355_KV_INT8 */
    int32_t page;       /* tokens per block: 16, 32 or 64 */
    int32_t nkv;        /* local kv heads */
    int32_t hd;         /* head dim: 64 or 128 */
    int32_t num_blocks;
    int32_t act_dtype;  /* dtype of Q / K / V rows and of the attention output: MI355_ACT_F16 / MI355_ACT_BF16.  A 16-bit cache holds
                         * that dtype (kv_dtype MI355_KV_FP16 <-> F16, MI355_KV_BF16 <-> BF16); the INT8 cache pairs with either */
} mi355_kv_layer_t;

/*
 * Bias + RoPE + Q-extract + paged KV write for decode — replaces
 * FusedRopeKVCacheDecodeOp::forward (bindings/rocm/FusedRopeKVCacheOp.cc:519-646,
 * kernel fused_rope_kvcache_kernel.cu:1297-1466).  NeoX pairing (i, i+rope_dim/2),
 * fp32 rotation with the fp32 {cos,sin} table cos_sin[pos][rope_dim/2][2]
 * (RopeCache.cc:16-41, interleave=true form), one rounding to fp16.
 * Input is either fp16 qkv[T][(nh+2nkv)*hd] or nsplit fp32 slabs of the QKV GEMM.
 * Token t is written at position positions[t] of sequence t (decode: one token
 * per sequence) through block_table[t][pos / page].
 * INT8 cache: scale = max|x| / 127 per (token, kv head) fp32, q = rne_sat(x / scale)
 * (rounding of rocm_utils/_cast_to_int8.h:5-24); the byte stored is q + 128 (offset-binary, private to this writer and
 * mi355_paged_*attn: the pool owner never interprets cache bytes).
 * Range checks (device side; the reference's kernel has none): a token whose position is outside
 * [0, min(max_pos, max_blocks_per_seq * page)) or whose block id is outside [0, kv->num_blocks) is NOT written to the
 * cache (its q row is still produced, rotated with the clamped position) and *oob_count (dev, may be NULL) is
 * incremented once per such token, so a stale position cannot overwrite another sequence's pages.
 */
int mi355_rope_kv_write(const void* qkv_f16, const float* partials, int32_t nsplit, int32_t ld,
                        const void* qkv_bias, const float* cos_sin, int32_t rope_dim, int32_t max_pos,
                        const int32_t* positions, const int32_t* block_table,
                        int32_t max_blocks_per_seq, int32_t T, int32_t nh,
                        const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count, mi355_stream_t stream);

/* Fused fast paths of the decode step (gemm_fullk.hip): one launch, no split-K workspace.  All return
 * MI355_ERR_UNSUPPORTED (nothing launched) for shapes / formats they do not take -- W4 group-wise and fp16 weights with
 * K a multiple of 128 and M <= 64 are taken (with a fused norm: M <= 16); the caller then composes mi355_linear_forward +
 * the separate op.
 *
 * mi355_fused_norm_t: RMSNorm applied to the GEMM's input on the fly, x_n = weight * fp16(h * rs) (the arithmetic of
 *   mi355_add_rmsnorm; reference modules/base/common/norm.py:83-92), with rs = rsqrt(sum_k h^2 / K + eps) rebuilt from
 *   per-tile partial sums: tile_sumsq[row * ld + t] = sum over the 16 columns of tile t of h[row]^2, tiles = K / 16 (a
 *   multiple of 4), ld >= tiles and a multiple of 4, 16-byte aligned -- what
 *   mi355_linear_residual leaves behind for the rows it produces.  The norm launch between two GEMMs disappears.
 * mi355_linear_residual: residual_out = residual_in + fp16(x W + bias)   (may alias residual_in); tile_sumsq_out
 *   (nullable) receives the per-tile sums of residual_out^2, [M][tile_sumsq_ld >= N / 16].
 *   replaces o_proj / down_proj followed by the residual add of the reference decoder layer
 *   (rtp_llm/models_py/model_desc/qwen3.py:63-77: hidden_states = residual + hidden_states, twice per layer).
 * mi355_norm_linear: y = epilogue(RMSNorm(h) W + bias): post-attention norm + gate_up + SiLU-gate in one launch
 *   (qwen3.py:74-76 + modules/hybrid/dense_mlp.py:95-106); epilogue as mi355_linear_forward.
 * mi355_qkv_rope_kv_write: [RMSNorm +] QKV projection + bias + NeoX RoPE + Q extract + fp16 paged KV write in one launch
 *   replaces LinearBase.forward (linear_base.py:75-85) followed by FusedRopeKVCacheDecodeOp::forward
 *   (FusedRopeKVCacheOp.cc:519-646); arguments as mi355_rope_kv_write_rows; norm == NULL: x holds the normed rows. */
typedef struct {
    const float* tile_sumsq; /* [rows][ld] fp32 */
    int32_t      tiles, ld;
    const void*  weight;     /* fp16 [K] */
    float        eps;
} mi355_fused_norm_t;

int mi355_linear_residual(const void* x, int32_t M, const mi355_weight_t* w, const void* bias, const void* residual_in,
                          void* residual_out, float* tile_sumsq_out, int32_t tile_sumsq_ld, mi355_stream_t stream);
int mi355_norm_linear(const void* h, int32_t M, const mi355_fused_norm_t* norm, const mi355_weight_t* w, const void* bias,
                      void* y, int32_t epilogue, mi355_stream_t stream);
int mi355_qkv_rope_kv_write(const void* x, int32_t M, const mi355_weight_t* wqkv, const void* qkv_bias,
                            const mi355_fused_norm_t* norm, const float* cos_sin, int32_t rope_dim, int32_t max_pos,
                            const int32_t* positions, const int32_t* block_table, int32_t max_blocks_per_seq, int32_t q_len,
                            int32_t nh, const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count, mi355_stream_t stream);
/*
 * 5-64-row steps: the fused launches above with the activations handed over as an IMAGE (gemm_fullk64.hip).
 * With K inside one block, every block pulls all M activation rows of its K range through its CU; gathered from the row-major tensor
 * that is 16 runs of 64 B per MFMA fragment at the row stride, which the L2s serve at a fifth of the rate of dense 1 KB runs when the
 * whole chip asks for the same rows (profiles/r04_fullk64_access_patterns.txt: QKV at 64 rows 19.8 vs 11.2 us).  The image stores
 * x[M][K] (16-bit elements) fragment by fragment: element (row, col) at index
 *   ((((col / 32) * ceil(M / 16) + row / 16) * 64 + ((col % 32) / 8) * 16 + row % 16) * 8) + col % 8
 * i.e. 16-byte pieces of a row stay whole, so a producer's vector store only changes its address.  Rows past M inside the last row
 * block are never read back as results.  Producers: mi355_add_rmsnorm_img (y), mi355_paged_attn_rows_img (out), or
 * mi355_act_image_pack from a row-major tensor (direction 1: back).  Consumers: mi355_qkv_rope_kv_write_img,
 * mi355_linear_residual_img: same arguments and results as the entry points without the suffix (no fused norm: the producer
 * normed), W4 group-wise weights (K <= 5760; up to 9600 where the launch splits its rows over two blocks per tile pair: a TP shard) or per-channel W8
 * (load-time INT8 autoquant, device_impl.py:183-222; K <= 3840), 1 <= M <= 64 (an instance per row-block count; the step driver takes them from 5 rows,
 * from 1 for bf16 / W8 / under tensor parallelism), else MI355_ERR_UNSUPPORTED.
 * Reference boundary these keep: modules/hybrid/causal_attention.py:75-93 (qkv_proj -> rope / kv write -> attention -> o_proj),
 * model_desc/qwen3.py:57-79 (norm -> attention -> residual add).
 */
/* Deferred RMSNorm between the O projection and gate_up of a 5-64-row step (no norm launch): the producer stores, besides the new
 * residual rows h, the image of g = fp16(weight * 2^-norm_exp * h) and the per-tile sums of h^2 (as mi355_linear_residual does);
 * the consumer runs its GEMM on g and multiplies the accumulators by unscale * rsqrt(sum_k h^2 / K + eps), unscale = 2^norm_exp:
 * (RMSNorm(h) W)[m][n] = rs[m] * sum_k weight[k] h[m][k] W[k][n].  norm_exp >= log2(max |weight|) keeps |g| <= |h|: no fp16
 * overflow whatever the residual stream holds.  One rounding of the activations to fp16 instead of the reference's two
 * (modules/base/common/norm.py:83-92: fp16(h * rs), then * weight), the row factor in fp32. */
typedef struct {
    const float* tile_sumsq; /* [rows][ld]: per-tile sums of h^2, tiles = K / 16 of them per row */
    int32_t      tiles, ld;
    float        eps;
    float        unscale;    /* 2^norm_exp */
} mi355_deferred_norm_t;
size_t mi355_act_image_bytes(int32_t M, int32_t K);
/* direction 0: row-major [M][K] tensor of act_dtype -> image; 1: image -> row-major tensor of act_dtype.  An image always holds fp16
 * (the GEMMs that read it run fp16 MFMAs).  The image of a BF16 tensor holds fp16(x * 2^-8), saturated at +-65504: bf16 activations up to
 * 1.6e7 (the SiLU * up product of a bf16 checkpoint) pass through, every consumer launch scales its fp32 accumulators by 2^8 (it knows the
 * dtype from mi355_weight_t.act_dtype); exact for |x| >= 2^-6, an absolute spacing of 2^-16 below.  So an image is only meaningful together
 * with the dtype of the tensor it stands for. */
int mi355_act_image_pack(const void* src, int32_t M, int32_t K, void* dst, int32_t direction, int32_t act_dtype, mi355_stream_t stream);
int mi355_add_rmsnorm_img(const void* x, const float* partials, int32_t nsplit, int32_t ld, const void* bias,
                          const void* residual_in, void* residual_out, const void* weight, float eps, int32_t M, int32_t H,
                          void* y_img, int32_t act_dtype, mi355_stream_t stream);
int mi355_paged_attn_rows_img(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table,
                              int32_t max_blocks_per_seq, const int32_t* positions, int32_t B, int32_t q_len, int32_t nh,
                              float scale, int32_t max_seq_len, void* out_img, void* workspace, size_t workspace_bytes,
                              mi355_stream_t stream);
int mi355_linear_residual_img(const void* x_img, int32_t M, const mi355_weight_t* w, const void* bias, const void* residual_in,
                              void* residual_out, float* tile_sumsq_out, int32_t tile_sumsq_ld, mi355_stream_t stream);
/* mi355_linear_residual_img that also leaves the deferred-norm operands of the rows it produces: xg_img_out = image of
 * fp16(norm_weight * 2^-norm_exp * residual_out), tile_sumsq_out as above (both required) */
int mi355_linear_residual_prenorm_img(const void* x_img, int32_t M, const mi355_weight_t* w, const void* bias, const void* residual_in,
                                      void* residual_out, const void* norm_weight, int32_t norm_exp, void* xg_img_out,
                                      float* tile_sumsq_out, int32_t tile_sumsq_ld, mi355_stream_t stream);
/* y = epilogue(rs * (xg W) + bias): the wide GEMM (W4 group-wise, 1 <= M <= 64, N wide enough to fill the chip: gate_up) on an
 * activation image, with the deferred norm applied to the accumulators (dn may be null: a plain linear on an image) */
int mi355_linear_deferred_norm_img(const void* xg_img, int32_t M, const mi355_deferred_norm_t* dn, const mi355_weight_t* w,
                                   const void* bias, void* y, int32_t epilogue, mi355_stream_t stream);
/* y = epilogue(x W + bias) in ONE launch from an activation image for a linear whose N does NOT fill the chip in the wide GEMM's form -- a
 * column-parallel shard under tensor parallelism (gate_up of Qwen2-7B at tp 2 / 4, of Llama-3-70B at tp 8: DenseMLP.gate_up_proj,
 * modules/hybrid/dense_mlp.py:95-103, split by ffn_sp_neg1, utils/model_weight.py:265-277): 2-5 tiles per block, all of K inside the block's
 * 8 K-slice waves, fused SiLU-gate / image output (MI355_EPI_SILU_MUL, MI355_EPI_OUT_IMAGE).  W4 g128, 1 <= M <= 64; MI355_ERR_UNSUPPORTED when
 * no tile count puts 160-256 blocks on the chip (the caller uses mi355_linear_forward / the wide GEMM). */
int mi355_linear_direct_img(const void* x_img, int32_t M, const mi355_weight_t* w, const void* bias, void* y, int32_t epilogue,
                            mi355_stream_t stream);
/* fp32 split-K slabs [n][M][N_pad] of a deep-K linear (down_proj) from an activation image, for mi355_add_rmsnorm(_img) to fold:
 * returns n (<= max_splits, <= 16), or MI355_ERR_UNSUPPORTED (not W4 group-wise / more than 64 rows / K too short or too deep for
 * 8 K-slice waves of <= 5 chunks per block: the caller uses mi355_linear_partial on the row-major tensor) */
int mi355_linear_partial_img(const void* x_img, int32_t M, const mi355_weight_t* w, float* partials, int32_t max_splits,
                             mi355_stream_t stream);
int mi355_qkv_rope_kv_write_img(const void* x_img, int32_t M, const mi355_weight_t* wqkv, const void* qkv_bias,
                                const float* cos_sin, int32_t rope_dim, int32_t max_pos, const int32_t* positions,
                                const int32_t* block_table, int32_t max_blocks_per_seq, int32_t q_len, int32_t nh,
                                const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count, mi355_stream_t stream);

/* The same for q_len rows per sequence (speculative verify, chunked prefill): token t = row t % q_len of sequence t / q_len,
 * block_table is [T / q_len][max_blocks_per_seq]; positions[t] < 0 marks a padding row (q produced, nothing stored). */
int mi355_rope_kv_write_rows(const void* qkv_f16, const float* partials, int32_t nsplit, int32_t ld,
                             const void* qkv_bias, const float* cos_sin, int32_t rope_dim, int32_t max_pos,
                             const int32_t* positions, const int32_t* block_table,
                             int32_t max_blocks_per_seq, int32_t T, int32_t q_len, int32_t nh,
                             const mi355_kv_layer_t* kv, void* q_out, int32_t* oob_count, mi355_stream_t stream);

/*
 * Paged decode attention (flash-decoding, split over the sequence) — replaces
 * AiterDecodeAttnOp*.forward / paged_attention_atrex
 * (factory/attention/rocm_impl/aiter.py:1340-1561, bindings/rocm/atrexPA.cc:444-496).
 * out[B][nh*hd] = softmax(scale * q.K^T) . V over seq_lens[b] tokens gathered
 * through block_table[b][*]; fp32 logits/softmax; GQA group nh/nkv <= 16.
 * seq_lens[b] is clamped to [1, max_seq_len] and block ids to [0, kv->num_blocks) on the device: out-of-range inputs give
 * a wrong row, never an out-of-bounds access.
 */
size_t mi355_paged_attn_workspace_bytes(int32_t B, int32_t nh, int32_t hd, int32_t max_seq_len);

int mi355_paged_decode_attn(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table,
                            int32_t max_blocks_per_seq, const int32_t* seq_lens, int32_t B,
                            int32_t nh, float scale, int32_t max_seq_len, void* out,
                            void* workspace, size_t workspace_bytes, mi355_stream_t stream);

/*
 * q_len > 1 query rows per sequence against the paged cache, causal inside the page walk: row i of sequence b (index
 * b * q_len + i into q / positions / out) attends the tokens 0 .. positions[b * q_len + i] of sequence b, which must be
 * in the cache (mi355_rope_kv_write of the same rows runs first).  positions < 0 marks a padding row (output zeros).
 * The KV of a sequence is streamed once per 2 * (16 / group) rows.  Serves the speculative target-verify step
 * (`is_target_verify`, bindings/OpDefs.h:283: gamma + 1 rows per sequence) and chunked prefill over the paged cache
 * (FusedRopeKVCacheOp.cc:216-461 + the paged prefill ops of factory/attention/rocm_impl/aiter.py:244-950).
 * workspace: mi355_paged_attn_workspace_bytes(B * q_len, nh, hd, max_seq_len).
 */
int mi355_paged_attn_rows(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table,
                          int32_t max_blocks_per_seq, const int32_t* positions, int32_t B, int32_t q_len, int32_t nh,
                          float scale, int32_t max_seq_len, void* out, void* workspace, size_t workspace_bytes,
                          mi355_stream_t stream);

/* Greedy fast path: ids[b] = argmax(logits[b, :]) on fp32 logits, lowest index on ties
 * (bindings/core/CudaSampleOp.cc:687-700).  workspace >= B * 64 * 8 bytes. */
int mi355_argmax(const float* logits, int32_t B, int32_t V, int32_t ld, int32_t* ids,
                 void* workspace, size_t workspace_bytes, mi355_stream_t stream);

/* Speculative decoding, target-verify step.
 * mi355_rejection_sample: chain rejection sampling of gamma draft tokens per row against the target model, argument
 * for argument invokeRejectionSampling<float,int>
 * (rtp_llm/models_py/bindings/rocm/speculative_sampling/sampling.cu:306-530):
 *   draft_probs   [B][gamma][V] fp32 (ignored when draft_probs_point_mass != 0: p = 1 at the draft token)
 *   draft_token_ids [B][gamma], uniform_samples [B][gamma+1] in [0,1)
 *   target_probs  [B][gamma+1][V] fp32, target_token_ids [B][gamma+1][stride] (the LAST element of each record is used)
 *   do_sample     [B] bytes: 0 = greedy row (accept iff draft == target token), 1 = accept iff u * p < q, on rejection
 *                 draw from relu(q - p) with the next uniform (first index whose prefix sum exceeds u * sum)
 *   output_token_ids [B][gamma+1]: accepted drafts, then the correction / bonus token, then -1 padding
 *   output_accepted_token_num [B]: accepted drafts + 1.
 * mi355_softmax_rows: probs[r][:] = softmax(logits[r][:] / temperature) in fp32 (the rows the sampler hands over). */
int mi355_softmax_rows(const float* logits, int32_t rows, int32_t V, int32_t ld, float temperature, float* probs,
                       mi355_stream_t stream);
/* ids[r] = first index whose inclusive fp32 prefix sum of probs[r, :] (index order) exceeds uniform_samples[r] * sum --
 * sampling from the probabilities, the top_k = 0 / top_p = 1 branch of the sampler (bindings/core/CudaSampleOp.cc:702-737);
 * gives the target model's own sampled token per verify row (target_token_ids of mi355_rejection_sample). */
int mi355_sample_rows(const float* probs, int32_t rows, int32_t V, int32_t ld, const float* uniform_samples, int32_t* ids,
                      mi355_stream_t stream);
/* The non-greedy branch of the sampler (sampleGreedy, bindings/core/CudaSampleOp.cc:619-800), on fp32 logit rows.
 * mi355_apply_penalties: in place, one launch (+ one memset node):
 *   temperature [B] or NULL: logit *= 1 / (T + 1e-6)          (batchApplyTemperaturePenalty, sampling_penalty_kernels.cu:26-54;
 *                the caller passes NULL when every T == 1, as CudaSampleOp.cc:633-645 skips the launch)
 *   repetition / presence / frequency [B] or NULL, output_ids [step][batch_size] int32 (the transposed token history),
 *   input_lengths [B] or NULL, max_input_length, step: argument for argument invokeBatchApplyRepetitionPenalty
 *   (sampling_penalty_kernels.cu:129-213): every id seen in the history (padding [input_length, max_input_length) skipped)
 *   is penalised once: logit = logit < 0 ? logit * rep : logit / rep; logit -= presence; logit -= frequency * count.
 *   penalty_ws: [B][V] int32 scratch (zeroed here).
 * mi355_top_k_top_p_sample (CudaSampleOp.cc:748-786): per row of probabilities, drop what is below the top_k-th largest value
 *   (top_k <= 0 or >= V: keep all; ties with the k-th value stay), drop an entry when the mass before it in descending order
 *   (equal values in index order) exceeds top_p (|top_p| < 1e-7 reads as 1), renormalise by max(sum, 1e-10) into probs_out (optional, may alias probs) and draw
 *   ids[r] by inverse CDF in index order with uniform_samples[r] (the reference calls torch.multinomial: same distribution).
 *   top_k / top_p may be NULL (no filter of that kind). */
int mi355_apply_penalties(float* logits, int32_t batch_size, int32_t V, int32_t ld, const float* temperature,
                          const float* repetition_penalty, const float* presence_penalty, const float* frequency_penalty,
                          const int32_t* output_ids, const int32_t* input_lengths, int32_t max_input_length, int32_t step,
                          int32_t* penalty_ws, mi355_stream_t stream);
/* no_repeat_ngram_size (invokeBanRepeatNgram, bindings/common/kernels/banRepeatNgram.cu:30-170, as called for greedy rows from
 * CudaSampleOp.cc:242-283): token_ids [batch][token_ld] int32 (row b = the tokens of sequence b so far), sequence_last_index [batch]
 * (index of the last valid token: the kernel's N = index + 1), no_repeat_ngram_size [batch] (0 = off).  Every earlier occurrence of
 * the sequence's last n - 1 tokens bans the token that followed it: logits[b][that token] = -inf. */
int mi355_ban_repeat_ngram(float* logits, int32_t batch_size, int32_t V, int32_t ld, const int32_t* token_ids, int32_t token_ld,
                           const int32_t* sequence_last_index, const int32_t* no_repeat_ngram_size, mi355_stream_t stream);
int mi355_top_k_top_p_sample(const float* probs, int32_t rows, int32_t V, int32_t ld, const int32_t* top_k, const float* top_p,
                             const float* uniform_samples, int32_t* ids, float* probs_out, int32_t ld_out, mi355_stream_t stream);
int mi355_rejection_sample(const float* draft_probs, const int32_t* draft_token_ids, const float* uniform_samples,
                           const float* target_probs, const int32_t* target_token_ids, int32_t target_token_stride,
                           int32_t* output_token_ids, int32_t* output_accepted_token_num, const uint8_t* do_sample,
                           int32_t batch_size, int32_t num_speculative_tokens, int32_t vocab_size,
                           int32_t draft_probs_point_mass, mi355_stream_t stream);

/* ------------------------------------------------------------------------
 * Tensor-parallel all-reduce over peer-mapped memory (xGMI) -- replaces the reference's custom one-shot all-reduce
 * TrtllmArFusionHandle (bindings/rocm/TrtllmAllReduceFusion.h:14-55; kernels trtllm_allreduce_fusion.cu:431-540) and,
 * at the call sites all_reduce(x, Group.TP) of causal_attention.py:91-92 / dense_mlp.py:104-105, RCCL for the decode
 * message sizes.  One process per GPU: every rank creates a context (which allocates and IPC-exports its buffers and
 * fills an opaque handle blob), the host gathers the `world` blobs in rank order by any means (torch.distributed
 * all_gather_object, as base/rocm/trt_allreduce.py:51-230 does) and every rank opens them.  Numerics: fp32 sum of the
 * fp16 copies in rank order 0..N-1, one rounding -- bit-identical on every rank (trtllm_allreduce_fusion.cu:228-246).
 * Tensors of more than 64 rows (prefill chunks) on more than two ranks take the two-shot form inside the same call: rank r
 * reduces rows r, r + N, ... and every rank fetches each row once from its owner -- 2 (N - 1) / N instead of (N - 1) element
 * reads per element over the links, bit-identical results (trtllm_allreduce_fusion.cu:606-692).
 * All calls only enqueue one kernel: graph-capturable; epochs advance on the device.  A peer that never arrives makes the
 * kernel give up after ~2 s and sets a status word (mi355_allreduce_status != 0) instead of hanging the GPU.
 * ---------------------------------------------------------------------- */
typedef struct mi355_allreduce mi355_allreduce_t;

size_t             mi355_allreduce_handle_bytes(void);
/* max_bytes: largest fp16 message (T * H * 2).  handle_out: mi355_allreduce_handle_bytes() bytes. */
mi355_allreduce_t* mi355_allreduce_create(int32_t rank, int32_t world, size_t max_bytes, void* handle_out);
/* all_handles: world blobs, rank order. */
int                mi355_allreduce_open(mi355_allreduce_t* ar, const void* all_handles);
void               mi355_allreduce_destroy(mi355_allreduce_t* ar);
int                mi355_allreduce_status(mi355_allreduce_t* ar, mi355_stream_t stream); /* synchronises; 0 = healthy */
/* bound of every in-kernel wait for a peer (default 2000 ms); ranks sharing ONE device (single-GPU validation runs) are time-sliced
 * against each other and want a longer one.  Applies to launches enqueued or captured afterwards. */
int                mi355_allreduce_set_spin_timeout_ms(mi355_allreduce_t* ar, int32_t ms);

/* out[T, n * world] = the ranks' [T, n] column slices side by side, out[t][r n + j] = x_r[t][j]: the all-gather of the
 * reference's hidden-split embedding (modules/base/common/embedding.py:50-58: all_gather, then reshape(tp, m, n).transpose(0, 1)
 * .reshape(m, -1)).  Same transport, same graph-capture rules as the all-reduce; T * n * world * 2 <= max_bytes; out != x. */
int mi355_allgather_hidden(mi355_allreduce_t* ar, const void* x_f16, void* out_f16, int32_t T, int32_t n, mi355_stream_t stream);

/* out[T,H] = sum over ranks of x[T,H] (fp16).  out may alias x. */
int mi355_allreduce_sum(mi355_allreduce_t* ar, const void* x_f16, void* out_f16, int32_t T, int32_t H, mi355_stream_t stream);

/* The fused form (allreduce_fusion_kernel_1stage + the split-K reduce of the producing row-parallel GEMM):
 *   local   = fp16(x_f16 | sum of nsplit fp32 slabs [nsplit][T][ld] (+ bias on rank 0))
 *   s       = fp16(sum over ranks of local)                       (rank order, fp32)
 *   h       = residual_in ? fp16(s + residual_in) : s ;  residual_out = h (if non-NULL)
 *   y       = weight * fp16(h * rsqrt(mean(h^2) + eps))            (if y non-NULL)
 * i.e. bit for bit mi355_allreduce_sum followed by mi355_add_rmsnorm. */
int mi355_allreduce_fused(mi355_allreduce_t* ar, const void* x_f16, const float* partials, int32_t nsplit, int32_t ld,
                          const void* bias, const void* residual_in, void* residual_out, const void* weight, float eps,
                          int32_t T, int32_t H, void* y, mi355_stream_t stream);

/* Hand-over protocol of the context's launches: 0 (default) = the published rows are stored write-through at system scope (sc0 sc1) and
 * ordered in front of the flags by the wave's own s_waitcnt; 1 = plain stores between system-scope release / acquire fences (an L2
 * write-back + invalidate per block; rounds 1-4).  Results are identical; also settable with MI355_AR_FULL_FENCES=1 at creation. */
int mi355_allreduce_set_full_fences(mi355_allreduce_t* ar, int32_t on);
/* The hand-over as one setting.  1 (default): write-through publishing stores + flags; 0 (opt-in, also MI355_AR_LL=1 at creation): the <= 64-row one-shot
 * calls of a decode step publish DATA-TAGGED GRANULES -- every 4 payload bytes in an aligned 8-byte {payload, epoch} written by one write-through store; a
 * peer polls the granules themselves: no flag table, no block barrier, one fabric round trip per all-reduce (NCCL's LL protocol) -- everything else as 1;
 * 2: plain stores between system-scope release / acquire fences (rounds 1-4).  Results are bit-identical in all three.  The granule form is validated with
 * 2 and 4 processes on one GPU and is left off by default: what it saves is an xGMI hop no development box had. */
int mi355_allreduce_set_protocol(mi355_allreduce_t* ar, int32_t mode);
/* a spin that timed out leaves mi355_allreduce_status != 0 for good; clear it once the host has dealt with the cause */
int mi355_allreduce_clear_status(mi355_allreduce_t* ar, mi355_stream_t stream);

/* In-launch prefetch for the NEXT mi355_allreduce_fused[_dt] launch of this context (cleared by that launch): while a block waits
 * for its peers' flags its other waves touch one dword per 128-byte line of [ptr, ptr + bytes) -- normally the weight shard of the
 * GEMM that consumes the all-reduce -- so the HBM fetch runs under the xGMI exchange without a side stream (the reference's
 * enable_comm_overlap hook, ConfigModules.h:275-282, at kernel granularity).  A launch of T <= 256 blocks covers at most
 * T * 448 * 8 lines.  ptr NULL or bytes 0: off.  No effect on the results. */
int mi355_allreduce_set_prefetch(mi355_allreduce_t* ar, const void* ptr, size_t bytes);

/* the same two calls with the tensors in act_dtype (MI355_ACT_BF16: bf16 copies are exchanged, sums stay fp32 in rank order);
 * mi355_allgather_hidden copies 16-bit elements and mi355_allreduce_argmax works on fp32 logits: both serve either dtype */
int mi355_allreduce_sum_dt(mi355_allreduce_t* ar, const void* x, void* out, int32_t T, int32_t H, int32_t act_dtype,
                           mi355_stream_t stream);
int mi355_allreduce_fused_dt(mi355_allreduce_t* ar, const void* x, const float* partials, int32_t nsplit, int32_t ld,
                             const void* bias, const void* residual_in, void* residual_out, const void* weight, float eps,
                             int32_t T, int32_t H, void* y, int32_t act_dtype, mi355_stream_t stream);
/* mi355_allreduce_fused_dt with y written as an activation image (mi355_act_image_*; T <= 64, H % 32 == 0): under tensor parallelism the
 * all-reduce point behind down_proj (modules/hybrid/dense_mlp.py:104-105) feeds the next layer's QKV launch on images
 * (mi355_qkv_rope_kv_write_img), as mi355_add_rmsnorm_img does at tp = 1.  Same numbers, other addresses. */
int mi355_allreduce_fused_img_dt(mi355_allreduce_t* ar, const void* x, const float* partials, int32_t nsplit, int32_t ld,
                                 const void* bias, const void* residual_in, void* residual_out, const void* weight, float eps,
                                 int32_t T, int32_t H, void* y_img, int32_t act_dtype, mi355_stream_t stream);

/* Round 6 -- a row-parallel shard without split-K slabs or a publishing stage.  The reference's all-reduce reads its input in place when it already lies
 * in a registered range (trtllm_allreduce_fusion.cu, TrtllmArFusionHandle: "inputs are first staged into a registered workspace unless already in a
 * captured/registered range"); here the producing GEMM writes there directly:
 *   mi355_linear_publish_img            y = 16-bit(xW + bias) of a 1-64-row step (x an activation image, W4 g128 / per-channel W8, K <= 5760, or <= 9600 when
 *                                       N / 32 <= 128) as ONE full-K launch whose epilogue stores the rows into this rank's registered buffer, in the slot and
 *                                       parity the NEXT fused all-reduce call of `ar` reads (bias: pass it on rank 0 only);
 *   mi355_allreduce_fused_published_dt  that call: flag exchange, rank-order fp32 sum of the N ranks' rows, residual add, RMSNorm (y row-major, or an
 *                                       activation image when y_is_image) -- mi355_allreduce_fused[_img]_dt without its fold + publish stage.
 * Both on one stream, no other call on `ar` in between.  Reduce sites: modules/hybrid/causal_attention.py:91-92 (O), dense_mlp.py:104-105 (down).
 * MI355_ERR_UNSUPPORTED (shape / format / a context on the granule protocol): stay on mi355_linear_partial* + mi355_allreduce_fused*. */
int mi355_linear_publish_img(const void* x_img, int32_t M, const mi355_weight_t* w, const void* bias, mi355_allreduce_t* ar, mi355_stream_t stream);
int mi355_allreduce_fused_published_dt(mi355_allreduce_t* ar, const void* residual_in, void* residual_out, const void* weight, float eps,
                                       int32_t T, int32_t H, void* y, int32_t y_is_image, int32_t act_dtype, mi355_stream_t stream);

/* Greedy sampling under a vocab-split lm_head: ids[b] = argmax over ALL ranks' logit slices (this rank holds columns
 * [vocab_offset, vocab_offset + V_local)), lowest global index on ties; identical on every rank.  Exchanges 8 bytes per
 * row instead of gathering the logits (PyWrappedModel.cc:915-936).  positions (may be NULL) += 1.
 * workspace >= B * 64 * 8 bytes. */
int mi355_allreduce_argmax(mi355_allreduce_t* ar, const float* logits, int32_t B, int32_t V_local, int32_t ld,
                           int32_t vocab_offset, int32_t* ids, int32_t* positions, void* workspace, size_t workspace_bytes,
                           mi355_stream_t stream);

/* ------------------------------------------------------------------------
 * Decode-step driver (C++): owns no tensors, only pointers.  It enqueues the
 * whole decode step of a Qwen2/Llama-style decoder (the body of
 * Qwen3Model.forward, rtp_llm/models_py/model_desc/qwen3.py:57-79,124-138,
 * plus lm_head + greedy of PyWrappedModel.cc:938-1080) and can capture it into
 * a hipGraph per batch size (the job of rtp_llm/cpp/cuda_graph/cuda_graph_runner.cc).
 * With tp_size > 1 the step is cut at the two all-reduce points of each layer
 * (causal_attention.py:91-92, dense_mlp.py:104-105); the caller reduces
 * `ar_buf` between segments.
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t num_layers, hidden, nh, nkv, hd, inter, vocab; /* per-rank values */
    int32_t rope_dim, max_pos;
    float   rms_eps;
    int32_t kv_dtype, page, num_blocks;
    int32_t max_batch, max_blocks_per_seq, max_seq_len;
    int32_t tp_size;
    int32_t act_dtype;   /* MI355_ACT_F16 / MI355_ACT_BF16: dtype of the embedding table, norm weights, biases, the hidden / ar_buf step
                          * buffers and of every linear (each mi355_weight_t.act_dtype must agree); a 16-bit KV cache has the
                          * same dtype (MI355_KV_BF16 for bf16), the INT8 cache goes with either */
} mi355_model_config_t;

typedef struct {
    mi355_weight_t qkv, o, gate_up, down; /* gate_up: interleaved (gate,up) columns */
    const void*    qkv_bias;              /* fp16 [(nh+2nkv)*hd] or NULL */
    const void*    input_norm;            /* fp16 [hidden] */
    const void*    post_norm;             /* fp16 [hidden] */
    void*          kv_base;               /* this layer's cache */
    float*         kv_scale_base;
} mi355_layer_weights_t;

typedef struct {
    const void*    embedding;  /* fp16 [vocab_full, hidden] */
    int32_t        vocab_full;
    const void*    final_norm; /* fp16 [hidden] */
    mi355_weight_t lm_head;    /* W16/W8/W4, N = vocab (per rank) */
    const float*   cos_sin;    /* fp32 [max_pos][rope_dim/2][2] */
} mi355_model_weights_t;

/* Step I/O buffers (device, caller-owned, address-stable for graph replay) */
typedef struct {
    int32_t* token_ids;   /* [max_batch]   in: current token; out (greedy): next token */
    int32_t* positions;   /* [max_batch]   tokens already in cache (= sequence_lengths) */
    int32_t* block_table; /* [max_batch][max_blocks_per_seq] */
    float*   logits;      /* [max_batch][vocab] fp32 */
    void*    hidden;      /* fp16 [max_batch][hidden]: last hidden state (post final norm) */
    void*    ar_buf;      /* fp16 [max_batch][hidden]: tensor to all-reduce (tp_size > 1) */
    void*    workspace;
    size_t   workspace_bytes;
} mi355_step_buffers_t;

typedef struct mi355_decoder mi355_decoder_t;

size_t           mi355_decoder_workspace_bytes(const mi355_model_config_t* cfg);
mi355_decoder_t* mi355_decoder_create(const mi355_model_config_t* cfg,
                                      const mi355_layer_weights_t* layers,
                                      const mi355_model_weights_t* model,
                                      const mi355_step_buffers_t* bufs);
void             mi355_decoder_destroy(mi355_decoder_t* d);

/* Segments, in order, for batch B (all enqueue on `stream`):
 *   begin            : embedding -> residual stream
 *   layer_attn(l)    : norm, QKV, RoPE+KV write, attention, O-proj
 *                      (tp>1: reduced fp16 result left in ar_buf)
 *   layer_mlp(l)     : (+residual) norm, gate_up+SiLU, down (tp>1: ar_buf)
 *   finish           : final norm, lm_head -> logits, greedy argmax -> token_ids,
 *                      positions += 1
 * mi355_decoder_step = all of them (tp_size == 1 only). */
int mi355_decoder_begin(mi355_decoder_t* d, int32_t B, mi355_stream_t stream);
/* The same step over nseq sequences x q_len rows each (nseq * q_len <= max_batch): token_ids / positions hold
 * nseq * q_len entries (row i of sequence b at b * q_len + i, positions ascending, < 0 = padding row), block_table one row
 * per SEQUENCE.  Row i attends the cache up to its own position: the speculative target-verify step (is_target_verify,
 * bindings/OpDefs.h:283).  Follow with the layer segments and mi355_decoder_finish as usual. */
int mi355_decoder_begin_rows(mi355_decoder_t* d, int32_t nseq, int32_t q_len, mi355_stream_t stream);
int mi355_decoder_layer_attn(mi355_decoder_t* d, int32_t layer, mi355_stream_t stream);
int mi355_decoder_layer_mlp(mi355_decoder_t* d, int32_t layer, mi355_stream_t stream);
int mi355_decoder_finish(mi355_decoder_t* d, int32_t sample, mi355_stream_t stream);
int mi355_decoder_step(mi355_decoder_t* d, int32_t B, mi355_stream_t stream);

/* Prefill over the paged cache (reference: prefill half of FusedRopeKVCacheOp.cc:216-461 and the paged prefill
 * ops of factory/attention/rocm_impl/aiter.py:244-950): one chunk of nseq sequences x q_len consecutive prompt tokens each
 * (row i of sequence b at b * q_len + i; positions[...] = token position, < 0 = padding row of a ragged batch; block_table
 * [nseq][max_blocks_per_seq]).  K/V of the chunk are stored, every row attends the cache up to its own position (earlier
 * chunks of the prompt included), linears take the large-M kernel.  logit_rows [nseq] (may be NULL) selects one row per
 * sequence whose fp32 logits [nseq][vocab] are written to logits_out (the last prompt token on the final chunk).
 * All pointers are device pointers; nothing synchronises. */
size_t mi355_decoder_prefill_workspace_bytes(mi355_decoder_t* d, int32_t max_tokens, int32_t max_seqs);
int    mi355_decoder_prefill(mi355_decoder_t* d, const int32_t* token_ids, const int32_t* positions, const int32_t* block_table,
                             int32_t nseq, int32_t q_len, const int32_t* logit_rows, float* logits_out, void* workspace,
                             size_t workspace_bytes, mi355_stream_t stream);
/* Dynamic-NTK RoPE styles only: {cos, sin} rows [max_pos][rope_dim / 2][2] (fp32, device) the prefill chunks that follow rotate with, or NULL for the model's
 * table.  The reference's context_rope (bindings/common/kernels/rotary_position_embedding.h:1000-1025) gives every token of a prefill batch the base of the
 * batch's longest prompt (fused_rope_kvcache_kernel.cu:219-260), while decode uses the base of each position (the model's table). */
int    mi355_decoder_set_prefill_rope_table(mi355_decoder_t* d, const float* cos_sin);

/* hipGraph: capture one full step for batch B on an internal stream, then replay
 * `nsteps` times back-to-back on `stream` (greedy feedback stays on device). */
/* Attach an opened all-reduce context (tp_size > 1): the two all-reduce points of every layer then run inside the step as
 * mi355_allreduce_fused (split-K reduce + all-reduce + residual + next RMSNorm in one launch) and greedy sampling as
 * mi355_allreduce_argmax, so mi355_decoder_step / _capture / _replay drive the whole tensor-parallel step from C++ with
 * no host round trip per layer.  vocab_offset: first vocabulary column of this rank's lm_head slice. */
int mi355_decoder_attach_allreduce(mi355_decoder_t* d, mi355_allreduce_t* ar, int32_t vocab_offset);
/* External collective transport: the fallback when the peer mapping of mi355_allreduce_open is not available (IPC refused
 * across devices / containers).  The reference does the same under graph capture: its custom kernel when it applies, raw
 * ncclAllReduce on the capture stream otherwise (rtp_llm/models_py/distributed/rocm_rccl.py:511-572, dispatch
 * collective_torch.py:694-722).  Both callbacks enqueue on `stream` and must be capturable (RCCL's are); they return 0 on
 * success.  With a transport attached, mi355_decoder_step / _capture / _replay / _prefill run for tp_size > 1 without an
 * mi355_allreduce_t: each row-parallel linear is reduced locally (split-K fold -> fp16), summed in place over the ranks by
 * all_reduce_f16, then residual + RMSNorm as one launch; greedy sampling all-gathers one (max, global index) pair per row. */
typedef struct mi355_collective {
    void*   ctx;
    int   (*all_reduce_f16)(void* ctx, void* buf, size_t count, mi355_stream_t stream);   /* in-place SUM of `count` fp16 */
    int   (*all_reduce_bf16)(void* ctx, void* buf, size_t count, mi355_stream_t stream);  /* the same for bf16; may be NULL (then a
                                                                                            * bf16 decoder refuses the transport) */
    int   (*all_gather)(void* ctx, const void* send, void* recv, size_t bytes_per_rank, mi355_stream_t stream); /* recv = [world][bytes] */
    int32_t rank, world;
} mi355_collective_t;
int mi355_decoder_attach_collective(mi355_decoder_t* d, const mi355_collective_t* coll, int32_t vocab_offset);

/* RCCL behind mi355_collective_t.  librccl is resolved at run time (dlopen of lib_path, NULL = "librccl.so"; a process that
 * already loaded torch's copy passes that path and shares it), so libmi355_decode.so has no link-time dependency on it.
 * Rank 0 makes the unique id (mi355_rccl_unique_id_bytes() bytes), the host passes it to the other ranks (any control
 * plane: gloo, a file, the reference's TCPStore), every rank opens. */
typedef struct mi355_rccl mi355_rccl_t;
size_t        mi355_rccl_unique_id_bytes(void);
int           mi355_rccl_unique_id(const char* lib_path, void* id_out);
mi355_rccl_t* mi355_rccl_open(const char* lib_path, const void* unique_id, int32_t rank, int32_t world);
int           mi355_rccl_collective(mi355_rccl_t* r, mi355_collective_t* out);
void          mi355_rccl_close(mi355_rccl_t* r);

/* on != 0: the embedding table given at creation holds only this rank's hidden / tp_size columns ([vocab][hidden / tp], the
 * reference's hidden-split embedding weight, modules/base/common/embedding.py:22-59); every step then looks its slice up and
 * all-gathers the hidden dimension (mi355_allgather_hidden) instead of reading a replicated table.  After attach_allreduce. */
int mi355_decoder_set_embedding_split(mi355_decoder_t* d, int32_t on);
/* Weight prefetch one launch ahead (tp_size == 1; under tensor parallelism the all-reduce points prefetch on their own, see
 * mi355_decoder_attach_allreduce): while a latency-bound launch runs, a side stream pulls the weights of a later linear into
 * the 256 MB Infinity Cache, joined (event edge, captured into the step graph) right before that linear.  The hook the
 * reference keeps for this is DeviceResourceConfig{enable_comm_overlap, overlap_comm_type} (rtp_llm/cpp/config/ConfigModules.h:275-282).
 * mask: MI355_PF_* bits, 0 = off.  Default: MI355_PF_QKV_IN_FOLD only -- the side-stream bits stay off (profiles/r03_prefetch_sidestream_ab.txt:
 * every fork / join pair costs ~17 us of step time inside a hipGraph on this stack).  Invalidates captured graphs. */
enum {
    MI355_PF_QKV      = 1,   /* next layer's QKV, requested before the down GEMM is launched */
    MI355_PF_O        = 2,   /* O, requested behind the QKV GEMM (runs under RoPE / KV write and attention) */
    MI355_PF_GATE_UP  = 4,   /* gate_up (first 48 MB), requested behind the O GEMM (runs under reduce + RMSNorm) */
    MI355_PF_QKV_LATE = 16,  /* next layer's QKV, requested behind the down GEMM (runs under reduce + RMSNorm only) */
    MI355_PF_O_LATE   = 32,  /* O, requested behind the RoPE / KV-write launch (runs under attention only) */
    MI355_PF_TP_COMM  = 64,  /* tp_size > 1: the next linear's shard while the fused all-reduce launch runs (round 2's default; off since
                              * the round-3 A/B: a fork / join pair inside the step graph costs more than the prefetch saves) */
    MI355_PF_QKV_IN_FOLD = 256, /* tp_size == 1, 5-64-row steps: the slab-fold launch behind down_proj runs on B of the 256 CUs; its spare blocks
                              * read one dword per 128-byte line of the NEXT layer's QKV weights (unit u on the XCD that will run the QKV
                              * launch's block u), so that launch finds them in the Infinity Cache / L2.  No stream, no graph edge. */
    MI355_PF_TP_INLAUNCH = 128 /* tp_size > 1, attached all-reduce context: the same overlap WITHOUT a second stream or a graph edge -- the
                              * waves of the fused all-reduce launch that only wait for the peers' flags request the first 8 MB of the
                              * next linear's shard (mi355_allreduce_set_prefetch).  Off by default until an 8-GPU A/B exists: the
                              * peer reads of the reduction queue behind the requests (vmcnt returns in order), which costs on one
                              * node what it may save on a loaded fabric; results are identical either way. */
};
int mi355_decoder_set_weight_prefetch(mi355_decoder_t* d, int32_t mask);

int mi355_decoder_capture(mi355_decoder_t* d, int32_t B);
int mi355_decoder_replay(mi355_decoder_t* d, int32_t B, int32_t nsteps, mi355_stream_t stream);

/* Per-kernel-class timing of `nsteps` eager steps with hipEvents on `stream`
 * (bench.py's roofline.achieved).  out_ms[class] = summed milliseconds,
 * out_launches[class] = launches; classes: MI355_KC_*. Synchronises the stream. */
enum {
    MI355_KC_GEMM_QUANT = 0, /* W4/W8 linears (qkv, o, gate_up, down) */
    MI355_KC_GEMM_LMHEAD = 1,
    MI355_KC_ATTN       = 2, /* paged decode attention (+partition reduce) */
    MI355_KC_ROPE_KV    = 3,
    MI355_KC_NORM       = 4,
    MI355_KC_OTHER      = 5,
    MI355_KC_COMM       = 6, /* fused all-reduce launches (tp_size > 1) */
    MI355_KC_COUNT      = 7
};
int mi355_decoder_profile(mi355_decoder_t* d, int32_t B, int32_t nsteps, float* out_ms,
                          int32_t* out_launches, mi355_stream_t stream);

/* Number of tokens the step driver refused to write to the KV cache since creation because their position or block id
 * was out of range (see mi355_rope_kv_write).  Synchronises `stream`.  0 on a healthy run; < 0 = error. */
int64_t mi355_decoder_oob_count(mi355_decoder_t* d, mi355_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_DECODE_H */
