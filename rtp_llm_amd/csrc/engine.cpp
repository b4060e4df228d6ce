// Decode-step driver: enqueues one whole decode step of a Qwen2/Llama-style decoder
// from C++ and replays it as a hipGraph.
//
// Scope: the body of Qwen3Model.forward / Qwen3DecoderLayer.forward
// (rtp_llm/models_py/model_desc/qwen3.py:57-79,124-138), CausalAttention.forward
// (modules/hybrid/causal_attention.py:75-93), DenseMLP.forward
// (modules/hybrid/dense_mlp.py:95-106), lm_head + greedy
// (rtp_llm/cpp/models/PyWrappedModel.cc:938-1080, bindings/core/CudaSampleOp.cc:687-700)
// and the per-batch-size graph capture of rtp_llm/cpp/cuda_graph/cuda_graph_runner.cc.
//
// Launches per layer (tp = 1):
//   * 5-64 rows (bf16 and per-channel W8: 1-64), W4 group-wise or W8 per-channel weights, 16-bit cache (rounds 4-5): SIX launches, activations handed over as fragment-ordered images
//     (common.h act_img_index): QKV + bias + RoPE + KV write (gemm_fullk64.hip) -> paged attention (+ partition reduce) ->
//     O + residual, leaving gamma 2^-e h' as an image and the sums of squares -> gate_up + SiLU with the RMSNorm finished on its
//     accumulators (gemm_wide.hip) -> down as 4 K-quarters (gemm_splitk64.hip) -> fold: slabs + residual + RMSNorm -> image;
//   * up to 4 rows every GEMM is a full-K launch with its consumer fused (gemm_fullk.hip), 6 launches:
//     [RMSNorm on load +] QKV + bias + RoPE + KV write -> attention -> partition reduce -> O + residual (leaves the
//     per-tile sums of squares) -> RMSNorm on load + gate_up + SiLU-gate -> down + residual;
//   * other formats / TP (above 12 rows): QKV GEMM (split-K slabs) -> reduce+bias+RoPE+KV-write -> paged attention (+ partition
//     reduce) -> O GEMM (slabs) -> reduce+residual+RMSNorm -> gate_up GEMM with fused SiLU-gate -> down GEMM (slabs) ->
//     reduce+residual+RMSNorm (already the next layer's input norm).  No standalone reduce / add / activation kernels.
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <string.h>
#include <map>
#include <new>
#include <vector>
#include "internal.h"
#ifdef MI355_TUNING
extern int g_tune[16];   // gemm.hip: experiment switches of the tuning build
#define TUNE(i) g_tune[i]
#else
#define TUNE(i) 0
#endif

void mi355_set_error(const char* fmt, ...);

namespace {
constexpr int kMaxSplits = 16;
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int    imax(int a, int b) { return a > b ? a : b; }
} // namespace

struct mi355_decoder {
    mi355_model_config_t               cfg;
    std::vector<mi355_layer_weights_t> layers;
    mi355_model_weights_t              model;
    mi355_step_buffers_t               bufs;
    // carved workspace
    void *resid, *xn, *q_buf, *attn_out, *act, *attn_ws, *argmax_ws;
    void *xn_img, *attn_img;   // activation images (mi355_act_image_*) of the normed hidden rows / the attention output, 5-64-row steps
    int32_t* oob_count; // tokens refused by the KV writer (stale position / block id)
    const float* prefill_cos_sin = nullptr;   // mi355_decoder_set_prefill_rope_table: {cos, sin} rows of the NEXT prefill chunks (dynamic-NTK prompts past the original context), or null
    mi355_allreduce_t* ar; // attached all-reduce context (tp_size > 1): the TP step runs entirely from C++
    int    vocab_offset;
    // external transport (RCCL) for the same points when the peer mapping is not available: local fold -> fp16 ar_buf ->
    // in-place all-reduce -> residual + norm launch; greedy = all-gather of one (max, global index) pair per row
    mi355_collective_t ext;
    bool   has_ext;
    void  *pairs_local, *pairs_all;
    bool   embed_split;   // the embedding table holds this rank's hidden / tp columns: lookup + all-gather (embedding.py:50-58)
    // comm / weight-stream overlap: while the (latency-bound, <= 64 blocks) all-reduce kernel runs on the main stream, a
    // side stream pulls the NEXT GEMM's weight shard into the Infinity Cache
    hipStream_t side_stream;
    hipEvent_t  ev_fork, ev_join;
    bool        overlap;
    // weight prefetch one launch ahead (all tp): while a latency- or issue-bound launch runs on the main stream, a side stream
    // streams the weights of a LATER linear into the Infinity Cache (256 MB, memory side); the linear then starts from cache
    // instead of paying the first-byte latency of HBM under its own launch burst.  pf_mask: MI355_PF_* bits.
    int         pf_mask;
    bool        pf_pending;
    float* partials;
    size_t attn_ws_bytes, argmax_ws_bytes, partials_bytes;
    void*    wide_ws;        // max_batch > 64: buffers of the generic large-batch step (mi355_decoder_prefill with q_len = 1)
    size_t   wide_ws_bytes;
    int32_t* iota;           // [max_batch] 0, 1, 2, ...: "logits of every row"
    int    B;       // rows of the step in flight (= sequences * q_len)
    int    q_len;   // rows per sequence: 1 = decode, > 1 = target-verify / prefill chunk (causal over the paged cache)
    // fused full-K launches (gemm_fullk.hip), decided once from the weight formats: QKV + bias + RoPE + KV write in one
    // launch (fp16 cache), O / down projection + residual add without split-K slabs (tp = 1: under TP the partial sums
    // of the row-parallel linears meet in the all-reduce first).  Only up to fuse_rows rows: every block reads ALL
    // activation rows of its K range from L2, which is free at a few rows and 2x slower than the staged split-K kernels
    // at 64 (measured M = 64: qkv+rope 20.8 vs 9.0 + 4.9 us, o 18.6 vs 9.7 + slab fold; M = 1: 6.8 vs 7.3 + 3.1 us)
    bool   fuse_qkv, fuse_o, fuse_down, fuse_norm;
    int    fuse_rows;
    // 1-64 rows (tp = 1): QKV + bias + RoPE + KV write and O + residual as one full-K launch each (gemm_fullk64.hip), their
    // activations handed over as images by the producing launches (RMSNorm fold, attention): 7 launches per layer instead of 8 + no slabs
    bool   img_qkv, img_o;
    // tensor parallelism with the in-step all-reduce attached (round 5): the column-parallel QKV shard runs as the same image launch --
    // the fused all-reduce behind down_proj (and the first norm of the step) write the image it reads; the row-parallel O / down shards
    // keep their slab launches (their partial sums meet in the all-reduce launch either way)
    bool   tp_img_qkv;
    int    tp_fuse_rows;    // up to this many rows the TP step keeps the few-row QKV launch (row-major input)
    // ... and the row-parallel down_proj shard as K quarters (gemm_splitk64.hip) from the image the gate_up shard's SiLU epilogue writes:
    // 2-4 slabs into the fused all-reduce launch instead of up to 15
    bool   tp_img_down;
    // ... and the column-parallel gate_up shard as ONE launch from an image (gemm_splitk64.hip, direct form) instead of the staged split-K kernel + fold:
    // the fused all-reduce behind the O shard then writes its normed rows as that image (into xg_img: unused under TP otherwise)
    bool   tp_img_gate;
    // round 6: the row-parallel O / down shards as ONE full-K launch each whose epilogue writes the rank's rows straight into the registered all-reduce
    // buffer (mi355_linear_publish_img), consumed by mi355_allreduce_fused_published_dt: no slabs, no fold + publish stage in front of the flag exchange
    bool   tp_pub_o, tp_pub_down;
    // ... and the post-attention RMSNorm deferred into gate_up's accumulators (mi355_deferred_norm_t): the O launch leaves
    // gamma 2^-e h' as an image + the per-tile sums of h'^2, the wide GEMM applies rsqrt(mean h'^2 + eps) 2^e: 6 launches per layer
    bool   img_gate_up;
    // ... and down_proj as 4 K-quarters from the image gate_up's SiLU epilogue writes (gemm_splitk64.hip) instead of 15 slabs
    bool   img_down;
    void*  act_img;
    std::vector<int> post_norm_exp;   // per layer: e >= log2(max |post_norm weight|)
    void*  xg_img;                    // image of gamma 2^-e h'
    float* ssq64;                     // [64][hidden / 16 rounded up to 4]
    float* ssq;      // [16 rows][hidden / 16]: the producer GEMM's per-tile sums of h^2, consumed by the next GEMM's on-the-fly RMSNorm
    // graphs
    hipStream_t                    cap_stream;
    std::map<int, hipGraphExec_t>  graphs;
    // profiling
    bool                     prof_on;
    std::vector<hipEvent_t>  ev;
    std::vector<int>         ev_class;
    size_t                   ev_used;
};

namespace {

struct Carve {
    char*  base;
    size_t off;
    void*  take(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    }
};

struct PrefillBufs;
size_t carve_prefill(const mi355_model_config_t& c, int T, int nseq, void* base, PrefillBufs* out);

size_t carve_all(mi355_decoder* d, const mi355_model_config_t& c, void* base) {
    Carve cv{(char*)base, 0};
    const size_t MB_ = (size_t)(c.max_batch < 64 ? c.max_batch : 64);   // the fused small-batch path handles <= 64 rows per step
    const int qdim = c.nh * c.hd, qkvdim = (c.nh + 2 * c.nkv) * c.hd;
    const int npad = (imax(qkvdim, c.hidden) + 15) & ~15;
    void* resid    = cv.take(MB_ * c.hidden * 2);
    void* xn       = cv.take(MB_ * c.hidden * 2);
    void* q_buf    = cv.take(MB_ * qdim * 2);
    void* attn_out = cv.take(MB_ * qdim * 2);
    void* act      = cv.take(MB_ * c.inter * 2);
    // split-K slabs: qkv / o / down use up to kMaxSplits of [max_batch, npad]; a K-split gate_up (TP shards) up to 4 of [max_batch, 2 inter]
    const size_t pbytes = std::max((size_t)kMaxSplits * MB_ * npad * 4, (size_t)4 * MB_ * ((2 * c.inter + 15) & ~15) * 4);
    void* partials = cv.take(pbytes);
    const size_t aw = mi355_paged_attn_workspace_bytes((int)MB_, c.nh, c.hd, c.max_seq_len);
    void* attn_ws  = cv.take(aw);
    const size_t gw = (size_t)c.max_batch * 64 * 8;
    void* argmax_ws = cv.take(gw);
    void* oob = cv.take(256);
    void* ssq = cv.take((size_t)16 * ((c.hidden / 16 + 3) & ~3) * 4);   // per-tile sums of squares of the residual rows (fused norm, <= 16 rows)
    void* xn_img = cv.take(mi355_act_image_bytes(64, c.hidden));
    void* xg_img = cv.take(mi355_act_image_bytes(64, c.hidden));
    void* act_img = cv.take(mi355_act_image_bytes(64, c.inter));
    void* ssq64 = cv.take((size_t)64 * ((c.hidden / 16 + 3) & ~3) * 4);
    void* attn_img = cv.take(mi355_act_image_bytes(64, qdim));
    const size_t wide_bytes = c.max_batch > 64 ? carve_prefill(c, c.max_batch, c.max_batch, nullptr, nullptr) : 0;
    void* wide_ws = cv.take(wide_bytes);
    void* iota = cv.take((size_t)c.max_batch * 4);
    if (d) {
        d->ssq = (float*)ssq; d->xn_img = xn_img; d->attn_img = attn_img; d->xg_img = xg_img; d->ssq64 = (float*)ssq64; d->act_img = act_img;
        d->oob_count = (int32_t*)oob; d->wide_ws = wide_ws; d->wide_ws_bytes = wide_bytes; d->iota = (int32_t*)iota;
        d->resid = resid; d->xn = xn; d->q_buf = q_buf; d->attn_out = attn_out; d->act = act;
        d->partials = (float*)partials; d->partials_bytes = pbytes; d->attn_ws = attn_ws; d->attn_ws_bytes = aw;
        d->argmax_ws = argmax_ws; d->argmax_ws_bytes = gw;
    }
    return cv.off;
}

// run one op, optionally bracketed by events for the per-class profile
template <class F>
int run_op(mi355_decoder* d, int kclass, hipStream_t st, F&& f) {
    if (!d->prof_on) return f();
    if (d->ev_used + 2 > d->ev.size()) {
        for (int i = 0; i < 2; ++i) { hipEvent_t e; hipEventCreate(&e); d->ev.push_back(e); }
    }
    hipEventRecord(d->ev[d->ev_used], st);
    const int rc = f();
    hipEventRecord(d->ev[d->ev_used + 1], st);
    d->ev_class.push_back(kclass);
    d->ev_used += 2;
    return rc;
}

#define ADT (d->cfg.act_dtype)   /* activation dtype of every norm call below */
#define RUN(kclass, expr)                                                  \
    do {                                                                   \
        int rc__ = run_op(d, kclass, st, [&]() -> int { return (expr); }); \
        if (rc__ < 0) return rc__;                                         \
    } while (0)

mi355_kv_layer_t kv_of(const mi355_decoder* d, int l) {
    mi355_kv_layer_t kv;
    kv.kv_base = d->layers[l].kv_base; kv.scale_base = d->layers[l].kv_scale_base;
    kv.kv_dtype = d->cfg.kv_dtype; kv.page = d->cfg.page; kv.nkv = d->cfg.nkv; kv.hd = d->cfg.hd;
    kv.num_blocks = d->cfg.num_blocks;
    kv.act_dtype = d->cfg.act_dtype;
    return kv;
}

} // namespace

extern "C" size_t mi355_decoder_workspace_bytes(const mi355_model_config_t* cfg) {
    if (!cfg) return 0;
    return carve_all(nullptr, *cfg, nullptr);
}

extern "C" mi355_decoder_t* mi355_decoder_create(const mi355_model_config_t* cfg, const mi355_layer_weights_t* layers,
                                                 const mi355_model_weights_t* model, const mi355_step_buffers_t* bufs) {
    if (!cfg || !layers || !model || !bufs) { mi355_set_error("decoder_create: null argument"); return nullptr; }
    if (cfg->num_layers <= 0 || cfg->max_batch <= 0 || cfg->max_batch > 4096) {
        mi355_set_error("decoder_create: num_layers=%d max_batch=%d (1..4096)", cfg->num_layers, cfg->max_batch);
        return nullptr;
    }
    if (cfg->nh % cfg->nkv != 0 || cfg->nh / cfg->nkv > 16 || (cfg->hd != 64 && cfg->hd != 128)) {
        mi355_set_error("decoder_create: nh=%d nkv=%d hd=%d unsupported", cfg->nh, cfg->nkv, cfg->hd);
        return nullptr;
    }
    if (cfg->max_seq_len <= 0 || cfg->max_seq_len > cfg->max_pos || cfg->page <= 0 ||
        (long)cfg->max_blocks_per_seq * cfg->page < cfg->max_seq_len || cfg->num_blocks <= 0) {
        mi355_set_error("decoder_create: max_seq_len=%d must be <= max_pos=%d (rotation table) and <= max_blocks_per_seq*page=%ld",
                        cfg->max_seq_len, cfg->max_pos, (long)cfg->max_blocks_per_seq * cfg->page);
        return nullptr;
    }
    {   // activation dtype: one for the whole step
        const bool bf = cfg->act_dtype == MI355_ACT_BF16;
        bool ok = (cfg->act_dtype == MI355_ACT_F16 || bf) && (cfg->kv_dtype == MI355_KV_INT8 || bf == (cfg->kv_dtype == MI355_KV_BF16));
        auto lin_ok = [&](const mi355_weight_t& w) { return w.act_dtype == cfg->act_dtype; };
        ok = ok && lin_ok(model->lm_head);
        for (int l = 0; ok && l < cfg->num_layers; ++l)
            ok = lin_ok(layers[l].qkv) && lin_ok(layers[l].o) && lin_ok(layers[l].gate_up) && lin_ok(layers[l].down);
        if (!ok) {
            mi355_set_error("decoder_create: act_dtype=%d needs every linear in that dtype%s and a %s KV cache", cfg->act_dtype,
                            "", bf ? "bf16 or INT8" : "fp16 or INT8");
            return nullptr;
        }
    }
    const size_t need = carve_all(nullptr, *cfg, nullptr);
    if (!bufs->workspace || bufs->workspace_bytes < need) {
        mi355_set_error("decoder_create: workspace %zu < %zu", bufs->workspace_bytes, need);
        return nullptr;
    }
    if (!bufs->token_ids || !bufs->positions || !bufs->block_table || !bufs->logits || !bufs->hidden ||
        (cfg->tp_size > 1 && !bufs->ar_buf)) {
        mi355_set_error("decoder_create: missing step buffer");
        return nullptr;
    }
    mi355_decoder* d = new (std::nothrow) mi355_decoder();
    if (!d) return nullptr;
    d->cfg = *cfg; d->layers.assign(layers, layers + cfg->num_layers); d->model = *model; d->bufs = *bufs;
    carve_all(d, *cfg, bufs->workspace);
    d->xn = bufs->hidden; // the normed hidden state lives in the caller-visible buffer
    d->B = 0; d->q_len = 1; d->cap_stream = nullptr; d->prof_on = false; d->ev_used = 0; d->ar = nullptr; d->vocab_offset = 0;
    d->has_ext = false; d->pairs_local = d->pairs_all = nullptr;
    d->side_stream = nullptr; d->ev_fork = d->ev_join = nullptr; d->overlap = false;
    d->pf_mask = MI355_PF_QKV_IN_FOLD; d->pf_pending = false;   // the one prefetch that needs no stream and measured positive (profiles/r04_touch_prefetch.txt)
    const bool bf_act = cfg->act_dtype == MI355_ACT_BF16;
    // QKV + RoPE + KV write in one launch: a 16-bit cache of the activation dtype (INT8 caches keep the quantising writer of rope_kv.hip)
    d->fuse_qkv = cfg->kv_dtype == (bf_act ? MI355_KV_BF16 : MI355_KV_FP16) && cfg->rope_dim == cfg->hd;
    d->fuse_o = d->fuse_down = cfg->tp_size == 1;
    // crossover between the few-row full-K launches (gemm_fullk.hip, dense activation loads) and the launches on activation images
    // (round 4, same box: b = 4 1.94 vs 2.05 ms, b = 5 2.16 vs 2.05, b = 8 2.28 vs 2.07, b = 12 2.64 vs 2.10; profiles/r04_batch_sweep_crossover.txt);
    // without the image path (other weight formats, bf16, TP) the round-3 crossover to the staged kernels stays
    d->fuse_rows = 12;
   // tuning build: crossover experiments (tools/batch_sweep.py --tune 6=N)
    // W4 layers only: for small fp16 models (the 0.5B draft of speculative decoding: 36-56 blocks per launch) the fused
    // launches measured behind the staged kernels (draft step 1.11 vs 1.05 ms), although the kernels take fp16 weights
    auto w4ok = [](const mi355_weight_t* w) { return w->wbits == 4 && mi355_fullk_weight_ok(w); };
    for (const auto& L : d->layers) {
        d->fuse_qkv = d->fuse_qkv && w4ok(&L.qkv);
        d->fuse_o = d->fuse_o && w4ok(&L.o);
        d->fuse_down = d->fuse_down && w4ok(&L.down);
    }
    // the norm launches disappear as well when every GEMM of the small-batch layer is a full-K launch: O / down leave the
    // per-tile sums of squares of the new residual rows, QKV / gate_up rebuild 1 / rms from them and normalise on load
    d->fuse_norm = d->fuse_qkv && d->fuse_o && d->fuse_down && cfg->hidden % 64 == 0;
    for (const auto& L : d->layers) d->fuse_norm = d->fuse_norm && w4ok(&L.gate_up);
    // the 17-64-row full-K launches: W4 group-wise, K <= 5760 (gemm_fullk64.hip), single rank
    // (either activation dtype: the images are fp16, the epilogues store the step's dtype -- the few-row launches above are fp16 only,
    // mi355_fullk_weight_ok)
    auto w64ok = [&](const mi355_weight_t* w) { return mi355_fullk64_weight_ok(w) != 0; };   // the launchers' own predicate (gemm.hip)
    d->img_qkv = cfg->kv_dtype == (bf_act ? MI355_KV_BF16 : MI355_KV_FP16) && cfg->rope_dim == cfg->hd && cfg->tp_size == 1;
    d->img_o = cfg->tp_size == 1;
    for (const auto& L : d->layers) { d->img_qkv = d->img_qkv && w64ok(&L.qkv); d->img_o = d->img_o && w64ok(&L.o); }
    d->tp_img_qkv = cfg->tp_size > 1 && cfg->kv_dtype == (bf_act ? MI355_KV_BF16 : MI355_KV_FP16) && cfg->rope_dim == cfg->hd && cfg->hidden % 32 == 0 &&
                    cfg->max_batch >= 1 && TUNE(5) != 2;
    for (const auto& L : d->layers) d->tp_img_qkv = d->tp_img_qkv && mi355_fullk64_qkv_ok(&L.qkv, cfg->hd) != 0;
    d->tp_img_down = cfg->tp_size > 1 && cfg->inter % 32 == 0 && TUNE(5) != 4 && TUNE(5) != 2;
    for (const auto& L : d->layers)
        d->tp_img_down = d->tp_img_down && L.down.K % 128 == 0 && L.down.K_pad == L.down.K && L.down.K == cfg->inter && L.gate_up.N == 2 * cfg->inter &&
                         mi355_gemm_splitk64_plan(64, L.down.N_pad / 16, L.down.K_pad / 128, L.down.wbits, L.down.group_size, kMaxSplits, nullptr) > 0;
    d->tp_img_gate = d->tp_img_down && cfg->hidden % 32 == 0 && TUNE(5) != 3;
    for (const auto& L : d->layers)
        d->tp_img_gate = d->tp_img_gate && L.gate_up.K % 128 == 0 && L.gate_up.K_pad == L.gate_up.K &&
                         mi355_gemm_splitk64_direct_plan(64, L.gate_up.N_pad / 16, L.gate_up.K_pad / 128, L.gate_up.wbits, L.gate_up.group_size) > 0;
    d->tp_pub_o = d->tp_img_qkv && !(TUNE(8) & 1);
    d->tp_pub_down = d->tp_img_down && !(TUNE(8) & 2);
    for (const auto& L : d->layers) {
        d->tp_pub_o = d->tp_pub_o && mi355_fullk64_publish_ok(&L.o) != 0;
        // down: where K stays inside two or three chunks per wave (K <= 5760: tp 4 of Qwen2-7B, tp 8 of the 70B models); deeper shards keep the K quarters
        d->tp_pub_down = d->tp_pub_down && mi355_fullk64_publish_ok(&L.down) != 0 && (L.down.K_pad / 128 <= 45 || (TUNE(8) & 4));
    }
    // from how many rows the TP step takes the image launches: with the down shard as K quarters they win from ONE row (one rank of tp 2,
    // b = 1 / 2 / 4: 1.67 / 1.68 / 1.72 ms against 1.75 / 1.77 / 1.82 with the few-row QKV launch + 15 staged slabs; profiles/r05_tp_small_batch_crossover.txt);
    // without that plan the few-row QKV launch keeps its rows (the tp = 1 crossover)
    d->tp_fuse_rows = (bf_act || !d->fuse_qkv || d->tp_img_down) ? 0 : 4;
    if (TUNE(6) > 0) d->tp_fuse_rows = TUNE(6) == 99 ? 0 : TUNE(6);   // tuning build: crossover experiments
    if (TUNE(5) == 2) d->img_qkv = d->img_o = false;   // tuning build: A/B against the split-K + fold launches
    // bf16 / W8: no few-row full-K launches to cross over to (gemm_fullk.hip takes fp16 steps of W4 / fp16 weights; the staged kernels lose
    // at every height): the image launches serve 1-64 rows
    if (d->img_qkv && d->img_o) d->fuse_rows = (bf_act || !(d->fuse_qkv && d->fuse_o && d->fuse_down)) ? 0 : 4;
    if (TUNE(6) > 0) d->fuse_rows = TUNE(6) == 99 ? 0 : TUNE(6);   // tuning build: crossover experiments (tools/batch_sweep.py --tune 6=N)
    d->img_gate_up = d->img_o && cfg->hidden % 64 == 0 && TUNE(5) != 3;
    for (const auto& L : d->layers) d->img_gate_up = d->img_gate_up && mi355_gemm_wide_direct_ok(&L.gate_up);
    d->img_down = d->img_gate_up && cfg->inter % 32 == 0 && TUNE(5) != 4;
    for (const auto& L : d->layers)
        d->img_down = d->img_down && L.down.K % 128 == 0 && L.down.K_pad == L.down.K &&
                      mi355_gemm_splitk64_plan(64, L.down.N_pad / 16, L.down.K_pad / 128, L.down.wbits, L.down.group_size, kMaxSplits, nullptr) > 0;
    // bf16: the wide GEMM's image entry only exists with an image OUTPUT (gemm.hip mi355_linear_deferred_norm_img): without the
    // K-quarter down launch that reads it, gate_up goes back to norm launch + staged kernel instead of failing every step
    if (bf_act && !d->img_down) d->img_gate_up = false;
    if (d->img_gate_up) {   // exponent of the deferred norm per layer: the largest |weight| of post_norm, read back once
        std::vector<uint16_t> g(cfg->hidden);
        for (const auto& L : d->layers) {
            if (hipMemcpy(g.data(), L.post_norm, (size_t)cfg->hidden * 2, hipMemcpyDeviceToHost) != hipSuccess) {
                mi355_set_error("decoder_create: cannot read the norm weights"); delete d; return nullptr;
            }
            float mx = 0.f;
            for (uint16_t b : g) {   // 16-bit pattern -> |value|
                float v;
                if (bf_act) { const uint32_t u = (uint32_t)(b & 0x7FFF) << 16; memcpy(&v, &u, 4); if (!(v == v)) v = INFINITY; }
                else {
                    const int e = (b >> 10) & 31, m = b & 1023;
                    v = e == 0 ? ldexpf((float)m, -24) : (e == 31 ? INFINITY : ldexpf((float)(m | 1024), e - 25));
                }
                if (v > mx) mx = v;
            }
            int ex = 0;
            while (ex < 14 && ldexpf(1.f, ex) < mx) ++ex;
            if (!(mx < INFINITY) || ldexpf(1.f, ex) < mx) { d->img_gate_up = false; break; }   // inf / nan / beyond 2^14 in a norm weight: keep the norm launch
            // a bf16 residual stream may exceed the fp16 range of the image: eight more binary orders of headroom (|h| < 1.6e7), paid for with
            // precision only on elements below 2^-6 of a typical one
            d->post_norm_exp.push_back(bf_act ? (ex + 8 > 14 ? 14 : ex + 8) : ex);
        }
    }
    std::vector<int32_t> iota_h(cfg->max_batch);
    for (int i = 0; i < cfg->max_batch; ++i) iota_h[i] = i;
    if (hipMemset(d->oob_count, 0, 256) != hipSuccess ||
        hipMemcpy(d->iota, iota_h.data(), iota_h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        mi355_set_error("decoder_create: cannot initialise the workspace"); delete d; return nullptr;
    }
    return d;
}

extern "C" void mi355_decoder_destroy(mi355_decoder_t* d) {
    if (!d) return;
    for (auto& kv : d->graphs) hipGraphExecDestroy(kv.second);
    if (d->cap_stream) hipStreamDestroy(d->cap_stream);
    if (d->side_stream) hipStreamDestroy(d->side_stream);
    if (d->ev_fork) hipEventDestroy(d->ev_fork);
    if (d->ev_join) hipEventDestroy(d->ev_join);
    for (auto e : d->ev) hipEventDestroy(e);
    if (d->pairs_local) hipFree(d->pairs_local);
    delete d;
}

extern "C" int mi355_decoder_begin(mi355_decoder_t* d, int32_t B, mi355_stream_t stream) {
    return mi355_decoder_begin_rows(d, B, 1, stream);
}

extern "C" int mi355_decoder_begin_rows(mi355_decoder_t* d, int32_t nseq, int32_t q_len, mi355_stream_t stream) {
    const int B = nseq * q_len;
    if (!d || nseq <= 0 || q_len <= 0 || B > d->cfg.max_batch || B > 64) {
        mi355_set_error("decoder_begin: %d sequences x %d rows exceed max_batch or the 64-row limit of the segmented step "
                        "(larger batches: mi355_decoder_step)", nseq, q_len);
        return MI355_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    d->B = B; d->q_len = q_len;
    const auto& c = d->cfg;
    if (d->embed_split) {   // hidden-split table: this rank's columns, then the ranks' slices side by side
        const int n = c.hidden / c.tp_size;
        RUN(MI355_KC_OTHER, mi355_embedding(d->bufs.token_ids, B, d->model.embedding, n, d->model.vocab_full, d->act, st));
        RUN(MI355_KC_COMM, mi355_allgather_hidden(d->ar, d->act, d->resid, B, n, st));
    } else {
        RUN(MI355_KC_OTHER, mi355_embedding(d->bufs.token_ids, B, d->model.embedding, c.hidden, d->model.vocab_full, d->resid, st));
    }
    if (c.tp_size == 1 || d->ar) {
        if ((d->img_qkv && B > d->fuse_rows) || (d->tp_img_qkv && d->ar && B > d->tp_fuse_rows))
            RUN(MI355_KC_NORM, mi355_add_rmsnorm_img(d->resid, nullptr, 0, 0, nullptr, nullptr, nullptr, d->layers[0].input_norm, c.rms_eps, B,
                                                     c.hidden, d->xn_img, ADT, st));
        else
            RUN(MI355_KC_NORM, mi355_rmsnorm_dt(d->resid, d->layers[0].input_norm, c.rms_eps, B, c.hidden, d->xn, ADT, st));
    }
    return MI355_OK;
}

extern "C" int mi355_decoder_attach_allreduce(mi355_decoder_t* d, mi355_allreduce_t* ar, int32_t vocab_offset) {
    if (!d || !ar || d->cfg.tp_size <= 1 || vocab_offset < 0) {
        mi355_set_error("decoder_attach_allreduce: needs a decoder created with tp_size > 1 and an opened context");
        return MI355_ERR_ARG;
    }
    for (auto& kv : d->graphs) hipGraphExecDestroy(kv.second);   // graphs captured for the segmented path are stale now
    d->graphs.clear();
    d->ar = ar; d->vocab_offset = vocab_offset;
    if (!d->side_stream) {
        d->overlap = hipStreamCreateWithFlags(&d->side_stream, hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming) == hipSuccess;
        (void)hipGetLastError();
    }
    return MI355_OK;
}

extern "C" int mi355_decoder_attach_collective(mi355_decoder_t* d, const mi355_collective_t* coll, int32_t vocab_offset) {
    if (!d || !coll || d->cfg.tp_size <= 1 || vocab_offset < 0 || !coll->all_reduce_f16 || !coll->all_gather ||
        coll->world != d->cfg.tp_size || coll->rank < 0 || coll->rank >= coll->world ||
        (d->cfg.act_dtype == MI355_ACT_BF16 && !coll->all_reduce_bf16)) {
        mi355_set_error("decoder_attach_collective: needs a decoder created with tp_size > 1 and a transport of that world size");
        return MI355_ERR_ARG;
    }
    if (d->ar) { mi355_set_error("decoder_attach_collective: an all-reduce context is already attached"); return MI355_ERR_ARG; }
    for (auto& kv : d->graphs) hipGraphExecDestroy(kv.second);
    d->graphs.clear();
    if (!d->pairs_local) {
        const size_t one = align256((size_t)d->cfg.max_batch * 8);
        if (hipMalloc(&d->pairs_local, one * (1 + (size_t)coll->world)) != hipSuccess) {
            mi355_set_error("decoder_attach_collective: %s", hipGetErrorString(hipGetLastError())); return MI355_ERR_HIP;
        }
        d->pairs_all = (char*)d->pairs_local + one;
    }
    d->ext = *coll; d->has_ext = true; d->vocab_offset = vocab_offset;
    return MI355_OK;
}

namespace {
// in-place sum of `count` fp16 over the TP ranks through the attached transport (no-op without one: the segment caller
// all-reduces ar_buf itself between the calls)
int ext_all_reduce(mi355_decoder* d, void* buf, size_t count, hipStream_t st) {
    if (!d->has_ext) return MI355_OK;
    const int rc = (d->cfg.act_dtype == MI355_ACT_BF16 ? d->ext.all_reduce_bf16 : d->ext.all_reduce_f16)(d->ext.ctx, buf, count, (mi355_stream_t)st);
    if (rc != 0) { mi355_set_error("decoder: the attached transport's all-reduce failed (%d)", rc); return MI355_ERR_HIP; }
    return MI355_OK;
}
} // namespace

extern "C" int mi355_decoder_set_embedding_split(mi355_decoder_t* d, int32_t on) {
    if (!d || (on && (!d->ar || d->cfg.tp_size <= 1 || d->cfg.hidden % (8 * d->cfg.tp_size) != 0))) {
        mi355_set_error("decoder_set_embedding_split: needs an attached all-reduce context and hidden %% (8 tp) == 0");
        return MI355_ERR_ARG;
    }
    for (auto& kv : d->graphs) hipGraphExecDestroy(kv.second);   // captured steps bake the embedding path in
    d->graphs.clear();
    d->embed_split = on != 0;
    return MI355_OK;
}

namespace {
constexpr size_t kPfInLaunchCap = (size_t)8 << 20;   // MI355_PF_TP_INLAUNCH: bytes of the next shard one all-reduce launch requests (the
                                                     // peer reads of its reduction queue behind them: vmcnt returns in order)
// all-reduce on `st`, prefetch of the next GEMM's weights on the side stream, joined before the GEMM (fork / join are
// plain event edges, so the pair is captured into the step graph like everything else)
template <class F>
int comm_with_prefetch(mi355_decoder* d, hipStream_t st, const mi355_weight_t* next_w, F&& comm) {
    const bool ov = d->overlap && (d->pf_mask & MI355_PF_TP_COMM) && next_w && next_w->qweight;
    if ((d->pf_mask & MI355_PF_TP_INLAUNCH) && d->ar && next_w && next_w->qweight) {
        // no second stream: the waves of the fused all-reduce launch that only wait for the peers' flags request the shard
        const size_t bytes = (size_t)next_w->K_pad * next_w->N_pad * next_w->wbits / 8;
        const int rc = mi355_allreduce_set_prefetch(d->ar, next_w->qweight, bytes < kPfInLaunchCap ? bytes : kPfInLaunchCap);
        if (rc < 0) return rc;
    }
    if (ov) {
        const size_t bytes = (size_t)next_w->K_pad * next_w->N_pad * next_w->wbits / 8;
        if (hipEventRecord(d->ev_fork, st) != hipSuccess || hipStreamWaitEvent(d->side_stream, d->ev_fork, 0) != hipSuccess) {
            mi355_set_error("decoder: fork to the side stream failed: %s", hipGetErrorString(hipGetLastError()));
            return MI355_ERR_HIP;
        }
        // the last dword of the oob block is never read by anyone: a harmless sink
        const int rc = mi355_prefetch(next_w->qweight, bytes < ((size_t)48 << 20) ? bytes : ((size_t)48 << 20), d->oob_count + 32, d->side_stream);
        if (rc < 0) return rc;
    }
    const int rc = comm();
    if (rc < 0) return rc;
    if (ov && (hipEventRecord(d->ev_join, d->side_stream) != hipSuccess || hipStreamWaitEvent(st, d->ev_join, 0) != hipSuccess)) {
        mi355_set_error("decoder: join of the side stream failed: %s", hipGetErrorString(hipGetLastError()));
        return MI355_ERR_HIP;
    }
    return rc;
}
} // namespace

namespace {
bool side_ready(mi355_decoder* d) {
    if (d->side_stream) return d->ev_fork && d->ev_join;
    const bool ok = hipStreamCreateWithFlags(&d->side_stream, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming) == hipSuccess;
    (void)hipGetLastError();
    return ok;
}
// fork: from this point of `st` on, the side stream pulls up to `cap` bytes of w into the Infinity Cache
int pf_issue(mi355_decoder* d, hipStream_t st, const mi355_weight_t* w, size_t cap) {
    if (!w || !w->qweight || d->pf_pending || !side_ready(d)) return MI355_OK;
    size_t bytes = (size_t)w->K_pad * w->N_pad * w->wbits / 8;
    if (bytes > cap) bytes = cap;
    if (hipEventRecord(d->ev_fork, st) != hipSuccess || hipStreamWaitEvent(d->side_stream, d->ev_fork, 0) != hipSuccess) {
        mi355_set_error("decoder: fork to the prefetch stream failed: %s", hipGetErrorString(hipGetLastError()));
        return MI355_ERR_HIP;
    }
    const int rc = mi355_prefetch(w->qweight, bytes, d->oob_count + 32, d->side_stream);
    if (rc < 0) return rc;
    if (hipEventRecord(d->ev_join, d->side_stream) != hipSuccess) {
        mi355_set_error("decoder: prefetch stream event failed: %s", hipGetErrorString(hipGetLastError()));
        return MI355_ERR_HIP;
    }
    d->pf_pending = true;
    return MI355_OK;
}
// join: launches issued on `st` after this point wait for the prefetch in flight (it has normally long finished)
int pf_join(mi355_decoder* d, hipStream_t st) {
    if (!d->pf_pending) return MI355_OK;
    d->pf_pending = false;
    if (hipStreamWaitEvent(st, d->ev_join, 0) != hipSuccess) {
        mi355_set_error("decoder: join of the prefetch stream failed: %s", hipGetErrorString(hipGetLastError()));
        return MI355_ERR_HIP;
    }
    return MI355_OK;
}
constexpr size_t kPfCap = (size_t)48 << 20;
} // namespace

extern "C" int mi355_decoder_set_weight_prefetch(mi355_decoder_t* d, int32_t mask) {
    if (!d || mask < 0) { mi355_set_error("decoder_set_weight_prefetch: bad argument"); return MI355_ERR_ARG; }
    for (auto& kv : d->graphs) hipGraphExecDestroy(kv.second);   // captured steps bake the fork / join edges in
    d->graphs.clear();
    if ((mask & ~(MI355_PF_TP_INLAUNCH | MI355_PF_QKV_IN_FOLD)) != 0 && !side_ready(d)) {   // the side stream and its events exist before any capture begins (not lazily inside one)
        mi355_set_error("decoder_set_weight_prefetch: cannot create the prefetch stream: %s", hipGetErrorString(hipGetLastError()));
        return MI355_ERR_HIP;
    }
    d->pf_mask = mask;
    return MI355_OK;
}

// does the attached context take published rows for a B-row call right now?  (host-side: not while it is on the granule protocol)
static bool pub_ready(const mi355_decoder_t* d, int B) {
    mi355_publish_target_t t;
    return d->ar && mi355_allreduce_publish_target(d->ar, B, d->cfg.hidden, &t) == MI355_OK;
}

extern "C" int mi355_decoder_layer_attn(mi355_decoder_t* d, int32_t l, mi355_stream_t stream) {
    if (!d || l < 0 || l >= d->cfg.num_layers || d->B <= 0) { mi355_set_error("decoder_layer_attn: layer=%d", l); return MI355_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const auto& c = d->cfg; const auto& L = d->layers[l]; const int B = d->B;
    if (c.tp_size > 1 && !d->ar) {
        if (l == 0) RUN(MI355_KC_NORM, mi355_rmsnorm_dt(d->resid, L.input_norm, c.rms_eps, B, c.hidden, d->xn, ADT, st));
        else RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(d->bufs.ar_buf, nullptr, 0, 0, nullptr, d->resid, d->resid, L.input_norm,
                                                   c.rms_eps, B, c.hidden, d->xn, ADT, st));
    }
    int ns = 0;
    mi355_kv_layer_t kv = kv_of(d, l);
    const bool tp_img = d->tp_img_qkv && d->ar && B > d->tp_fuse_rows;   // TP: the QKV shard on the image the all-reduce launch wrote
    const bool small = tp_img ? false : B <= d->fuse_rows;
    const bool normed = small && d->fuse_norm;            // no norm launches in this step: see fuse_norm
    const int pf = c.tp_size == 1 ? d->pf_mask : 0;       // (tp > 1: the side stream belongs to comm_with_prefetch)
    if (int e = pf_join(d, st)) return e;                  // this layer's QKV weights, requested behind the previous down GEMM
    const bool mid_qkv = (d->img_qkv && B > d->fuse_rows) || tp_img, mid_o = d->img_o && B > d->fuse_rows;   // 1-64 rows: full-K launches on activation images
    if (mid_qkv) {
        RUN(MI355_KC_GEMM_QUANT, mi355_qkv_rope_kv_write_img(d->xn_img, B, &L.qkv, L.qkv_bias, d->model.cos_sin, c.rope_dim, c.max_pos,
                                                             d->bufs.positions, d->bufs.block_table, c.max_blocks_per_seq, d->q_len, c.nh,
                                                             &kv, d->q_buf, d->oob_count, st));
        if (pf & (MI355_PF_O | MI355_PF_O_LATE)) if (int e = pf_issue(d, st, &L.o, kPfCap)) return e;
    } else if (d->fuse_qkv && small) {
        // layer 0 reads the rows mi355_decoder_begin normed; later layers normalise the residual rows on load
        const mi355_fused_norm_t fn = {d->ssq, c.hidden / 16, (c.hidden / 16 + 3) & ~3, L.input_norm, c.rms_eps};
        const bool on_load = normed && l > 0;
        RUN(MI355_KC_GEMM_QUANT, mi355_qkv_rope_kv_write(on_load ? d->resid : d->xn, B, &L.qkv, L.qkv_bias, on_load ? &fn : nullptr,
                                                         d->model.cos_sin, c.rope_dim, c.max_pos, d->bufs.positions,
                                                         d->bufs.block_table, c.max_blocks_per_seq, d->q_len, c.nh, &kv, d->q_buf,
                                                         d->oob_count, st));
        if (pf & (MI355_PF_O | MI355_PF_O_LATE)) if (int e = pf_issue(d, st, &L.o, kPfCap)) return e;
    } else {
        RUN(MI355_KC_GEMM_QUANT, ns = mi355_linear_partial(d->xn, B, &L.qkv, d->partials, kMaxSplits, st));
        if (pf & MI355_PF_O) if (int e = pf_issue(d, st, &L.o, kPfCap)) return e;          // under RoPE + attention
        RUN(MI355_KC_ROPE_KV, mi355_rope_kv_write_rows(nullptr, d->partials, ns, L.qkv.N_pad, L.qkv_bias, d->model.cos_sin, c.rope_dim,
                                                       c.max_pos, d->bufs.positions, d->bufs.block_table, c.max_blocks_per_seq, B,
                                                       d->q_len, c.nh, &kv, d->q_buf, d->oob_count, st));
        if (pf & MI355_PF_O_LATE) if (int e = pf_issue(d, st, &L.o, kPfCap)) return e;     // under attention only
    }
    // q_len > 1: rows of one sequence share a pass over its KV, causal mask inside the page walk (is_target_verify)
    if (tp_img && d->tp_pub_o && B <= 64 && pub_ready(d, B)) {   // TP: attention image -> O shard published in the all-reduce buffer -> exchange + residual + norm
        RUN(MI355_KC_ATTN, mi355_paged_attn_rows_img(d->q_buf, &kv, d->bufs.block_table, c.max_blocks_per_seq, d->bufs.positions,
                                                     B / d->q_len, d->q_len, c.nh, 1.0f / sqrtf((float)c.hd), c.max_seq_len, d->attn_img,
                                                     d->attn_ws, d->attn_ws_bytes, st));
        if (int e = pf_join(d, st)) return e;
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_publish_img(d->attn_img, B, &L.o, nullptr, d->ar, st));
        const bool gimg = d->tp_img_gate;                   // (B > tp_fuse_rows holds: tp_img)
        RUN(MI355_KC_COMM, comm_with_prefetch(d, st, &L.gate_up, [&]() {
            return mi355_allreduce_fused_published_dt(d->ar, d->resid, d->resid, L.post_norm, c.rms_eps, B, c.hidden, gimg ? d->xg_img : d->xn, gimg ? 1 : 0, ADT, st); }));
        return MI355_OK;
    }
    if (mid_o) {   // the attention output as an image: what the O projection's full-K launch reads
        RUN(MI355_KC_ATTN, mi355_paged_attn_rows_img(d->q_buf, &kv, d->bufs.block_table, c.max_blocks_per_seq, d->bufs.positions,
                                                     B / d->q_len, d->q_len, c.nh, 1.0f / sqrtf((float)c.hd), c.max_seq_len, d->attn_img,
                                                     d->attn_ws, d->attn_ws_bytes, st));
        if (int e = pf_join(d, st)) return e;
        if (d->img_gate_up) {   // no norm launch: gate_up finishes the RMSNorm on its accumulators
            RUN(MI355_KC_GEMM_QUANT, mi355_linear_residual_prenorm_img(d->attn_img, B, &L.o, nullptr, d->resid, d->resid, L.post_norm,
                                                                       d->post_norm_exp[l], d->xg_img, d->ssq64, (c.hidden / 16 + 3) & ~3, st));
            return MI355_OK;
        }
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_residual_img(d->attn_img, B, &L.o, nullptr, d->resid, d->resid, nullptr, 0, st));
        if (pf & MI355_PF_GATE_UP) if (int e = pf_issue(d, st, &L.gate_up, kPfCap)) return e;
        RUN(MI355_KC_NORM, mi355_rmsnorm_dt(d->resid, L.post_norm, c.rms_eps, B, c.hidden, d->xn, ADT, st));
        return MI355_OK;
    }
    RUN(MI355_KC_ATTN, mi355_paged_attn_rows(d->q_buf, &kv, d->bufs.block_table, c.max_blocks_per_seq, d->bufs.positions,
                                             B / d->q_len, d->q_len, c.nh, 1.0f / sqrtf((float)c.hd), c.max_seq_len, d->attn_out,
                                             d->attn_ws, d->attn_ws_bytes, st));
    if (int e = pf_join(d, st)) return e;
    if (d->fuse_o && small) {     // h += fp16(attn W_o) in the GEMM's epilogue: no slabs; the norm moves into gate_up (or stays a launch)
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_residual(d->attn_out, B, &L.o, nullptr, d->resid, d->resid, normed ? d->ssq : nullptr, (c.hidden / 16 + 3) & ~3, st));
        if (!normed) {
            if (pf & MI355_PF_GATE_UP) if (int e = pf_issue(d, st, &L.gate_up, kPfCap)) return e;
            RUN(MI355_KC_NORM, mi355_rmsnorm_dt(d->resid, L.post_norm, c.rms_eps, B, c.hidden, d->xn, ADT, st));
        }
        return MI355_OK;
    }
    RUN(MI355_KC_GEMM_QUANT, ns = mi355_linear_partial(d->attn_out, B, &L.o, d->partials, kMaxSplits, st));
    if (pf & MI355_PF_GATE_UP) if (int e = pf_issue(d, st, &L.gate_up, kPfCap)) return e;   // under the reduce + norm launch
    if (c.tp_size == 1) {
        RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(nullptr, d->partials, ns, L.o.N_pad, nullptr, d->resid, d->resid, L.post_norm,
                                             c.rms_eps, B, c.hidden, d->xn, ADT, st));
    } else if (d->ar) { // split-K reduce + all-reduce + residual + post-attention norm in one launch
        const bool gimg = d->tp_img_gate && B > d->tp_fuse_rows;   // the gate_up shard reads an image (one launch, gemm_splitk64 direct form)
        RUN(MI355_KC_COMM, comm_with_prefetch(d, st, &L.gate_up, [&]() {
            return gimg ? mi355_allreduce_fused_img_dt(d->ar, nullptr, d->partials, ns, L.o.N_pad, nullptr, d->resid, d->resid, L.post_norm,
                                                       c.rms_eps, B, c.hidden, d->xg_img, ADT, st)
                        : mi355_allreduce_fused_dt(d->ar, nullptr, d->partials, ns, L.o.N_pad, nullptr, d->resid, d->resid, L.post_norm,
                                         c.rms_eps, B, c.hidden, d->xn, ADT, st); }));
    } else { // local split-K reduce -> fp16 tensor for the TP all-reduce
        RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(nullptr, d->partials, ns, L.o.N_pad, nullptr, nullptr, d->bufs.ar_buf, nullptr,
                                             c.rms_eps, B, c.hidden, nullptr, ADT, st));
        RUN(MI355_KC_COMM, ext_all_reduce(d, d->bufs.ar_buf, (size_t)B * c.hidden, st));
    }
    return MI355_OK;
}

extern "C" int mi355_decoder_layer_mlp(mi355_decoder_t* d, int32_t l, mi355_stream_t stream) {
    if (!d || l < 0 || l >= d->cfg.num_layers || d->B <= 0) { mi355_set_error("decoder_layer_mlp: layer=%d", l); return MI355_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const auto& c = d->cfg; const auto& L = d->layers[l]; const int B = d->B;
    if (c.tp_size > 1 && !d->ar) {
        RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(d->bufs.ar_buf, nullptr, 0, 0, nullptr, d->resid, d->resid, L.post_norm, c.rms_eps,
                                             B, c.hidden, d->xn, ADT, st));
    }
    const bool small = B <= d->fuse_rows;
    const bool normed = small && d->fuse_norm && d->fuse_o;
    const int pf = c.tp_size == 1 ? d->pf_mask : 0;
    if (int e = pf_join(d, st)) return e;
    if (d->img_o && d->img_gate_up && B > d->fuse_rows) {   // the O launch left gamma 2^-e h' and the sums of h'^2: RMSNorm finished on gate_up's accumulators
        const mi355_deferred_norm_t dn = {d->ssq64, c.hidden / 16, (c.hidden / 16 + 3) & ~3, c.rms_eps, ldexpf(1.f, d->post_norm_exp[l])};
        const bool down_img = d->img_down;                 // gate_up's SiLU epilogue writes the image down_proj's K-quarter launch reads
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_deferred_norm_img(d->xg_img, B, &dn, &L.gate_up, nullptr, down_img ? d->act_img : d->act,
                                                                MI355_EPI_SILU_MUL | (down_img ? MI355_EPI_OUT_IMAGE : 0), st));
    } else if (normed) {   // post-attention RMSNorm on load + gate_up + SiLU-gate in one launch
        const mi355_fused_norm_t fn = {d->ssq, c.hidden / 16, (c.hidden / 16 + 3) & ~3, L.post_norm, c.rms_eps};
        RUN(MI355_KC_GEMM_QUANT, mi355_norm_linear(d->resid, B, &fn, &L.gate_up, nullptr, d->act, MI355_EPI_SILU_MUL, st));
    } else {
        const bool gi = d->tp_img_down && d->ar && B > d->tp_fuse_rows;   // TP: the SiLU output of the shard as the image the K-quarter down launch reads
        if (gi && d->tp_img_gate)   // ... and the shard itself as one launch on the image the all-reduce behind the O shard wrote
            RUN(MI355_KC_GEMM_QUANT, mi355_linear_direct_img(d->xg_img, B, &L.gate_up, nullptr, d->act_img, MI355_EPI_SILU_MUL | MI355_EPI_OUT_IMAGE, st));
        else
            RUN(MI355_KC_GEMM_QUANT, mi355_linear_direct(d->xn, B, &L.gate_up, nullptr, gi ? d->act_img : d->act, MI355_EPI_SILU_MUL | (gi ? MI355_EPI_OUT_IMAGE : 0),
                                                     d->partials, d->partials_bytes, st));
    }
    int ns = 0;
    const void* next_norm = (l + 1 < c.num_layers) ? d->layers[l + 1].input_norm : d->model.final_norm;
    const mi355_weight_t* next_qkv = (l + 1 < c.num_layers) ? &d->layers[l + 1].qkv : nullptr;
    if (pf & MI355_PF_QKV) if (int e = pf_issue(d, st, next_qkv, kPfCap)) return e;         // under the down GEMM (+ norm)
    if (d->fuse_down && small) {
        const bool last = l + 1 == c.num_layers;          // the final norm feeds lm_head: that one stays a launch
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_residual(d->act, B, &L.down, nullptr, d->resid, d->resid, (normed && !last) ? d->ssq : nullptr, (c.hidden / 16 + 3) & ~3, st));
        if (!normed || last) RUN(MI355_KC_NORM, mi355_rmsnorm_dt(d->resid, next_norm, c.rms_eps, B, c.hidden, d->xn, ADT, st));
        return MI355_OK;
    }
    if (d->tp_pub_down && d->ar && B > d->tp_fuse_rows && B <= 64 && pub_ready(d, B)) {   // TP: the down shard published in the all-reduce buffer (act_img: written above, gi)
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_publish_img(d->act_img, B, &L.down, nullptr, d->ar, st));
        const mi355_weight_t* next_w = (l + 1 < c.num_layers) ? &d->layers[l + 1].qkv : &d->model.lm_head;
        const bool next_img = d->tp_img_qkv && l + 1 < c.num_layers;
        RUN(MI355_KC_COMM, comm_with_prefetch(d, st, next_w, [&]() {
            return mi355_allreduce_fused_published_dt(d->ar, d->resid, d->resid, next_norm, c.rms_eps, B, c.hidden, next_img ? d->xn_img : d->xn, next_img ? 1 : 0, ADT, st); }));
        return MI355_OK;
    }
    if ((d->img_o && d->img_gate_up && d->img_down && B > d->fuse_rows) || (d->tp_img_down && d->ar && B > d->tp_fuse_rows))
        RUN(MI355_KC_GEMM_QUANT, ns = mi355_linear_partial_img(d->act_img, B, &L.down, d->partials, kMaxSplits, st));
    else
        RUN(MI355_KC_GEMM_QUANT, ns = mi355_linear_partial(d->act, B, &L.down, d->partials, kMaxSplits, st));
    if (pf & MI355_PF_QKV_LATE) if (int e = pf_issue(d, st, next_qkv, kPfCap)) return e;    // under the reduce + norm launch only
    if (c.tp_size == 1) {
        if (d->img_qkv && B > d->fuse_rows && l + 1 < c.num_layers)   // the next layer's QKV launch reads an image (the final norm feeds lm_head: row-major)
        {   // the fold runs on B of the 256 CUs: its spare blocks touch the NEXT layer's QKV weights (MI355_PF_QKV_IN_FOLD, internal.h mi355_touch_t)
            mi355_touch_t tc; const mi355_touch_t* touch = nullptr;
            if ((d->pf_mask & MI355_PF_QKV_IN_FOLD) && l + 1 < c.num_layers &&
                mi355_qkv_touch_plan(&d->layers[l + 1].qkv, c.hd, d->oob_count + 32, &tc) == MI355_OK) touch = &tc;
            RUN(MI355_KC_NORM, mi355_add_rmsnorm_img_touch(nullptr, d->partials, ns, L.down.N_pad, nullptr, d->resid, d->resid, next_norm,
                                                           c.rms_eps, B, c.hidden, d->xn_img, ADT, touch, st));
        }
        else
            RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(nullptr, d->partials, ns, L.down.N_pad, nullptr, d->resid, d->resid, next_norm,
                                                 c.rms_eps, B, c.hidden, d->xn, ADT, st));
    } else if (d->ar) {
        const mi355_weight_t* next_w = (l + 1 < c.num_layers) ? &d->layers[l + 1].qkv : &d->model.lm_head;
        const bool next_img = d->tp_img_qkv && B > d->tp_fuse_rows && l + 1 < c.num_layers;   // the next layer's QKV shard reads an image (the final norm feeds lm_head: row-major)
        RUN(MI355_KC_COMM, comm_with_prefetch(d, st, next_w, [&]() {
            return next_img ? mi355_allreduce_fused_img_dt(d->ar, nullptr, d->partials, ns, L.down.N_pad, nullptr, d->resid, d->resid, next_norm,
                                                           c.rms_eps, B, c.hidden, d->xn_img, ADT, st)
                            : mi355_allreduce_fused_dt(d->ar, nullptr, d->partials, ns, L.down.N_pad, nullptr, d->resid, d->resid, next_norm,
                                         c.rms_eps, B, c.hidden, d->xn, ADT, st); }));
    } else {
        RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(nullptr, d->partials, ns, L.down.N_pad, nullptr, nullptr, d->bufs.ar_buf, nullptr,
                                             c.rms_eps, B, c.hidden, nullptr, ADT, st));
        RUN(MI355_KC_COMM, ext_all_reduce(d, d->bufs.ar_buf, (size_t)B * c.hidden, st));
    }
    return MI355_OK;
}

// Dynamic-NTK RoPE (rotary_position_embedding.h:889-951) in PREFILL: context_rope (:1000-1025) rotates every token of the batch with ONE base, that of
// `seq_len` = the longest prompt of the batch (fused_rope_kvcache_kernel.cu:219-260 passes the kernel's padded sequence length), whereas the decode writer
// uses the base of each position -- which is what the model's table holds.  The host builds the {cos, sin} rows [max_pos][rope_dim / 2][2] for the batch's
// base and hands them over for the prefill chunks that follow; NULL returns to the model's table.  No effect on decode steps.
extern "C" int mi355_decoder_set_prefill_rope_table(mi355_decoder_t* d, const float* cos_sin) {
    if (!d) { mi355_set_error("decoder_set_prefill_rope_table: null decoder"); return MI355_ERR_ARG; }
    d->prefill_cos_sin = cos_sin;
    return MI355_OK;
}

// ------------------------------------------------------------------ prefill
// One chunk of nseq x q_len prompt tokens through the same weights: large-M GEMMs (gemm_prefill.hip above 128 rows), the
// rows-mode KV writer, causal multi-row attention over the paged cache (earlier chunks of the same prompt are already
// there), fused residual + RMSNorm; logits only for the requested rows (the reference computes lm_head on the last token
// of each sequence, PyWrappedModel.cc:1003-1047).
namespace {
struct PrefillBufs { void *resid, *xn, *qkv, *q, *attn, *act, *tmp, *gemm_ws, *attn_ws, *last_h; size_t gemm_ws_bytes, attn_ws_bytes; };

// upper bound of the split-K slab workspace mi355_linear_forward may ask for on 64-row slabs of this model's linears
size_t linear_ws_bound(const mi355_model_config_t& c) {
    const int qkvdim = (c.nh + 2 * c.nkv) * c.hd;
    const int shapes[4][2] = {{c.hidden, qkvdim}, {c.nh * c.hd, c.hidden}, {c.hidden, 2 * c.inter}, {c.inter, c.hidden}};
    size_t need = 256;
    for (const auto& sh : shapes)
        for (int wbits : {4, 8, 16}) {
            mi355_weight_t w;
            w.qweight = &need; w.meta = &need; w.wbits = wbits; w.K = sh[0]; w.N = sh[1]; w.act_dtype = c.act_dtype;   // bf16 plans other block shapes (and splits) above 32 rows
            w.K_pad = (sh[0] + 127) & ~127; w.N_pad = (sh[1] + 15) & ~15; w.group_size = wbits == 4 ? 128 : 0;
            need = std::max(need, mi355_linear_workspace_bytes(64, &w));
        }
    return need;
}

size_t carve_prefill(const mi355_model_config_t& c, int T, int nseq, void* base, PrefillBufs* out) {
    Carve cv{(char*)base, 0};
    const size_t Ts = (size_t)T;
    const int qdim = c.nh * c.hd, qkvdim = (c.nh + 2 * c.nkv) * c.hd;
    PrefillBufs b;
    b.resid = cv.take(Ts * c.hidden * 2); b.xn = cv.take(Ts * c.hidden * 2); b.qkv = cv.take(Ts * qkvdim * 2);
    b.q = cv.take(Ts * qdim * 2); b.attn = cv.take(Ts * qdim * 2); b.act = cv.take(Ts * c.inter * 2); b.tmp = cv.take(Ts * c.hidden * 2);
    b.last_h = cv.take((size_t)nseq * c.hidden * 2 * 2);
    // 64-row slabs through the decode kernels (chunks below 128 rows) may split K into fp32 slabs: ask the planner
    b.gemm_ws_bytes = linear_ws_bound(c);
    b.gemm_ws = cv.take(b.gemm_ws_bytes);
    b.attn_ws_bytes = mi355_paged_attn_workspace_bytes(T, c.nh, c.hd, c.max_seq_len);
    b.attn_ws = cv.take(b.attn_ws_bytes);
    if (out) *out = b;
    return cv.off;
}
} // namespace

extern "C" size_t mi355_decoder_prefill_workspace_bytes(mi355_decoder_t* d, int32_t max_tokens, int32_t max_seqs) {
    if (!d || max_tokens <= 0 || max_seqs <= 0) return 0;
    return carve_prefill(d->cfg, max_tokens, max_seqs, nullptr, nullptr);
}

extern "C" int mi355_decoder_prefill(mi355_decoder_t* d, const int32_t* token_ids, const int32_t* positions, const int32_t* block_table,
                                     int32_t nseq, int32_t q_len, const int32_t* logit_rows, float* logits_out, void* workspace,
                                     size_t workspace_bytes, mi355_stream_t stream) {
    if (!d || !token_ids || !positions || !block_table || nseq <= 0 || q_len <= 0 || !workspace) {
        mi355_set_error("decoder_prefill: bad arguments"); return MI355_ERR_ARG;
    }
    const auto& c = d->cfg;
    const int T = nseq * q_len;
    if (c.tp_size > 1 && !d->ar && !d->has_ext) {
        mi355_set_error("decoder_prefill: tp_size > 1 needs mi355_decoder_attach_allreduce or _attach_collective"); return MI355_ERR_ARG;
    }
    PrefillBufs b;
    const size_t need = carve_prefill(c, T, nseq, workspace, &b);
    if (need > workspace_bytes) { mi355_set_error("decoder_prefill: workspace %zu < %zu", workspace_bytes, need); return MI355_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)c.hd);
    if (d->embed_split) {
        const int n = c.hidden / c.tp_size;
        RUN(MI355_KC_OTHER, mi355_embedding(token_ids, T, d->model.embedding, n, d->model.vocab_full, b.tmp, st));
        RUN(MI355_KC_COMM, mi355_allgather_hidden(d->ar, b.tmp, b.resid, T, n, st));
    } else {
        RUN(MI355_KC_OTHER, mi355_embedding(token_ids, T, d->model.embedding, c.hidden, d->model.vocab_full, b.resid, st));
    }
    RUN(MI355_KC_NORM, mi355_rmsnorm_dt(b.resid, d->layers[0].input_norm, c.rms_eps, T, c.hidden, b.xn, ADT, st));
    for (int l = 0; l < c.num_layers; ++l) {
        const auto& L = d->layers[l];
        mi355_kv_layer_t kv = kv_of(d, l);
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_forward(b.xn, T, &L.qkv, L.qkv_bias, b.qkv, MI355_EPI_NONE, b.gemm_ws, b.gemm_ws_bytes, st));
        RUN(MI355_KC_ROPE_KV, mi355_rope_kv_write_rows(b.qkv, nullptr, 0, L.qkv.N, nullptr, d->prefill_cos_sin ? d->prefill_cos_sin : d->model.cos_sin, c.rope_dim, c.max_pos, positions,
                                                       block_table, c.max_blocks_per_seq, T, q_len, c.nh, &kv, b.q, d->oob_count, st));
        RUN(MI355_KC_ATTN, mi355_paged_attn_rows(b.q, &kv, block_table, c.max_blocks_per_seq, positions, nseq, q_len, c.nh, scale,
                                                 c.max_seq_len, b.attn, b.attn_ws, b.attn_ws_bytes, st));
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_forward(b.attn, T, &L.o, nullptr, b.tmp, MI355_EPI_NONE, b.gemm_ws, b.gemm_ws_bytes, st));
        if (c.tp_size > 1 && !d->ar) RUN(MI355_KC_COMM, ext_all_reduce(d, b.tmp, (size_t)T * c.hidden, st));
        if (c.tp_size == 1 || !d->ar) {
            RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(b.tmp, nullptr, 0, 0, nullptr, b.resid, b.resid, L.post_norm, c.rms_eps, T, c.hidden, b.xn, ADT, st));
        } else {
            RUN(MI355_KC_COMM, mi355_allreduce_fused_dt(d->ar, b.tmp, nullptr, 0, 0, nullptr, b.resid, b.resid, L.post_norm, c.rms_eps, T,
                                                     c.hidden, b.xn, ADT, st));
        }
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_forward(b.xn, T, &L.gate_up, nullptr, b.act, MI355_EPI_SILU_MUL, b.gemm_ws, b.gemm_ws_bytes, st));
        RUN(MI355_KC_GEMM_QUANT, mi355_linear_forward(b.act, T, &L.down, nullptr, b.tmp, MI355_EPI_NONE, b.gemm_ws, b.gemm_ws_bytes, st));
        const void* next_norm = (l + 1 < c.num_layers) ? d->layers[l + 1].input_norm : d->model.final_norm;
        if (c.tp_size > 1 && !d->ar) RUN(MI355_KC_COMM, ext_all_reduce(d, b.tmp, (size_t)T * c.hidden, st));
        if (c.tp_size == 1 || !d->ar) {
            RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(b.tmp, nullptr, 0, 0, nullptr, b.resid, b.resid, next_norm, c.rms_eps, T, c.hidden, b.xn, ADT, st));
        } else {
            RUN(MI355_KC_COMM, mi355_allreduce_fused_dt(d->ar, b.tmp, nullptr, 0, 0, nullptr, b.resid, b.resid, next_norm, c.rms_eps, T,
                                                     c.hidden, b.xn, ADT, st));
        }
    }
    if (logit_rows && logits_out) {   // gather the requested rows of the final normed hidden state, then lm_head on nseq rows
        RUN(MI355_KC_OTHER, mi355_embedding(logit_rows, nseq, b.xn, c.hidden, T, b.last_h, st));
        for (int r0 = 0; r0 < nseq; r0 += 64) {
            const int n = nseq - r0 < 64 ? nseq - r0 : 64;
            RUN(MI355_KC_GEMM_LMHEAD, mi355_linear_direct((const char*)b.last_h + (size_t)r0 * c.hidden * 2, n, &d->model.lm_head, nullptr,
                                                          logits_out + (size_t)r0 * c.vocab, MI355_EPI_OUT_F32, nullptr, 0, st));
        }
    }
    return MI355_OK;
}

namespace {
// vocab-split greedy through the external transport: (max, global index) per row, all-gathered, best per row
int ext_argmax(mi355_decoder* d, int B, hipStream_t st) {
    const auto& c = d->cfg;
    RUN(MI355_KC_OTHER, mi355_argmax_pairs(d->bufs.logits, B, c.vocab, c.vocab, d->vocab_offset, d->pairs_local, d->argmax_ws,
                                           d->argmax_ws_bytes, st));
    RUN(MI355_KC_COMM, d->ext.all_gather(d->ext.ctx, d->pairs_local, d->pairs_all, (size_t)B * 8, (mi355_stream_t)st) == 0
                           ? MI355_OK : (mi355_set_error("decoder: the attached transport's all-gather failed"), MI355_ERR_HIP));
    RUN(MI355_KC_OTHER, mi355_argmax_pick(d->pairs_all, d->ext.world, B, d->bufs.token_ids, d->bufs.positions, st));
    return MI355_OK;
}
} // namespace

extern "C" int mi355_decoder_finish(mi355_decoder_t* d, int32_t sample, mi355_stream_t stream) {
    if (!d || d->B <= 0) { mi355_set_error("decoder_finish: no step in flight"); return MI355_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const auto& c = d->cfg; const int B = d->B;
    if (int e = pf_join(d, st)) return e;
    if (c.tp_size > 1 && !d->ar) {
        RUN(MI355_KC_NORM, mi355_add_rmsnorm_dt(d->bufs.ar_buf, nullptr, 0, 0, nullptr, d->resid, d->resid, d->model.final_norm,
                                             c.rms_eps, B, c.hidden, d->xn, ADT, st));
    }
    RUN(MI355_KC_GEMM_LMHEAD, mi355_linear_direct(d->xn, B, &d->model.lm_head, nullptr, d->bufs.logits, MI355_EPI_OUT_F32, nullptr, 0, st));
    if (sample && d->ar) {   // vocab-split lm_head: (max, index) pairs cross the ranks, not the logits
        RUN(MI355_KC_COMM, mi355_allreduce_argmax(d->ar, d->bufs.logits, B, c.vocab, c.vocab, d->vocab_offset, d->bufs.token_ids,
                                                  d->bufs.positions, d->argmax_ws, d->argmax_ws_bytes, st));
    } else if (sample && d->has_ext) {
        if (int e = ext_argmax(d, B, st)) return e;
    } else if (sample) {
        RUN(MI355_KC_OTHER, mi355_argmax_ex(d->bufs.logits, B, c.vocab, c.vocab, d->bufs.token_ids, d->bufs.positions,
                                            d->argmax_ws, d->argmax_ws_bytes, st));
    }
    return MI355_OK;
}

extern "C" int mi355_decoder_step(mi355_decoder_t* d, int32_t B, mi355_stream_t stream) {
    if (!d || (d->cfg.tp_size != 1 && !d->ar && !d->has_ext)) {
        mi355_set_error("decoder_step: tp_size > 1 needs mi355_decoder_attach_allreduce / _attach_collective (or the segment calls)");
        return MI355_ERR_ARG;
    }
    if (B > 64) {   // large batch: the generic path (large-M GEMMs, fp16 intermediates), one row per sequence
        if (B > d->cfg.max_batch) { mi355_set_error("decoder_step: B=%d > max_batch", B); return MI355_ERR_ARG; }
        hipStream_t st = (hipStream_t)stream;
        const auto& c = d->cfg;
        int rc = mi355_decoder_prefill(d, d->bufs.token_ids, d->bufs.positions, d->bufs.block_table, B, 1, d->iota, d->bufs.logits,
                                       d->wide_ws, d->wide_ws_bytes, stream);
        if (rc < 0) return rc;
        if (d->ar) {
            RUN(MI355_KC_COMM, mi355_allreduce_argmax(d->ar, d->bufs.logits, B, c.vocab, c.vocab, d->vocab_offset, d->bufs.token_ids,
                                                      d->bufs.positions, d->argmax_ws, d->argmax_ws_bytes, st));
        } else if (d->has_ext) {
            if (int e = ext_argmax(d, B, st)) return e;
        } else {
            RUN(MI355_KC_OTHER, mi355_argmax_ex(d->bufs.logits, B, c.vocab, c.vocab, d->bufs.token_ids, d->bufs.positions, d->argmax_ws,
                                                d->argmax_ws_bytes, st));
        }
        return MI355_OK;
    }
    int rc = mi355_decoder_begin(d, B, stream);
    for (int l = 0; rc >= 0 && l < d->cfg.num_layers; ++l) {
        rc = mi355_decoder_layer_attn(d, l, stream);
        if (rc >= 0) rc = mi355_decoder_layer_mlp(d, l, stream);
    }
    if (rc >= 0) rc = mi355_decoder_finish(d, 1, stream);
    return rc < 0 ? rc : MI355_OK;
}

extern "C" int mi355_decoder_capture(mi355_decoder_t* d, int32_t B) {
    if (!d || B <= 0 || B > d->cfg.max_batch) { mi355_set_error("decoder_capture: B=%d", B); return MI355_ERR_ARG; }
    if (d->graphs.count(B)) return MI355_OK;
    hipError_t e = hipSuccess;
    if (!d->cap_stream && (e = hipStreamCreateWithFlags(&d->cap_stream, hipStreamNonBlocking)) != hipSuccess) {
        mi355_set_error("decoder_capture: stream create failed: %s", hipGetErrorString(e)); return MI355_ERR_HIP;
    }
    e = hipStreamBeginCapture(d->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { mi355_set_error("decoder_capture: begin: %s", hipGetErrorString(e)); return MI355_ERR_HIP; }
    const int rc = mi355_decoder_step(d, B, d->cap_stream);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(d->cap_stream, &g);
    if (rc < 0) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) { mi355_set_error("decoder_capture: end: %s", hipGetErrorString(e)); return MI355_ERR_HIP; }
    hipGraphExec_t ge = nullptr;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) { mi355_set_error("decoder_capture: instantiate: %s", hipGetErrorString(e)); return MI355_ERR_HIP; }
    d->graphs[B] = ge;
    return MI355_OK;
}

// Host-side helper for drivers that capture the TP step (collectives included) with their framework's graph API: if that
// capture is invalidated half way (an un-capturable collective), the stream is left capturing and every later launch
// fails.  Ends whatever capture is active on `stream`, discarding the partial graph.  Not part of the public ABI.
extern "C" int mi355_abort_capture(mi355_stream_t stream) {
    (void)hipGetLastError();   // the status query itself fails on an invalidated capture: just try to end it
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture((hipStream_t)stream, &g);
    if (g) hipGraphDestroy(g);
    (void)hipGetLastError();
    return MI355_OK;
}

extern "C" int mi355_decoder_replay(mi355_decoder_t* d, int32_t B, int32_t nsteps, mi355_stream_t stream) {
    if (!d || !d->graphs.count(B)) { mi355_set_error("decoder_replay: no graph for B=%d", B); return MI355_ERR_ARG; }
    hipGraphExec_t ge = d->graphs[B];
    for (int i = 0; i < nsteps; ++i) {
        hipError_t e = hipGraphLaunch(ge, (hipStream_t)stream);
        if (e != hipSuccess) { mi355_set_error("decoder_replay: %s", hipGetErrorString(e)); return MI355_ERR_HIP; }
    }
    return MI355_OK;
}

extern "C" int64_t mi355_decoder_oob_count(mi355_decoder_t* d, mi355_stream_t stream) {
    if (!d) { mi355_set_error("decoder_oob_count: null decoder"); return MI355_ERR_ARG; }
    int32_t v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
        hipMemcpy(&v, d->oob_count, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) {
        mi355_set_error("decoder_oob_count: %s", hipGetErrorString(hipGetLastError()));
        return MI355_ERR_HIP;
    }
    return v;
}

extern "C" int mi355_decoder_profile(mi355_decoder_t* d, int32_t B, int32_t nsteps, float* out_ms, int32_t* out_launches,
                                     mi355_stream_t stream) {
    if (!d || !out_ms || !out_launches || nsteps <= 0) { mi355_set_error("decoder_profile: bad args"); return MI355_ERR_ARG; }
    for (int k = 0; k < MI355_KC_COUNT; ++k) { out_ms[k] = 0.f; out_launches[k] = 0; }
    d->prof_on = true; d->ev_used = 0; d->ev_class.clear();
    int rc = MI355_OK;
    for (int i = 0; i < nsteps && rc >= 0; ++i) rc = mi355_decoder_step(d, B, stream);
    d->prof_on = false;
    hipStreamSynchronize((hipStream_t)stream);
    if (rc < 0) return rc;
    for (size_t i = 0; i < d->ev_class.size(); ++i) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, d->ev[2 * i], d->ev[2 * i + 1]);
        out_ms[d->ev_class[i]] += ms;
        out_launches[d->ev_class[i]] += 1;
    }
    return MI355_OK;
}
