// Native registration shim: the pybind11 / torch-extension face of libmi355_decode.so.
//
// The reference links exactly one `void rtp_llm::registerPyModuleOps(pybind11::module&)` per build flavour
// (rtp_llm/models_py/bindings/RegisterOps.h:9; ROCm flavour: bindings/rocm/RegisterRocmOps.cc:7-10 ->
// RegisterBaseBindings.hpp:14-129 + RegisterAttnOpBindings.hpp:8-11) and calls it from
// PYBIND11_MODULE(librtp_compute_ops, m) (rtp_llm/cpp/pybind/ComputeInit.cc:18-27); Python then reaches the ops as
// rtp_llm.ops.compute_ops.rtp_llm_ops.<name>.  This file defines that same hook for the MI355X flavour on top of the
// C-ABI (include/mi355_decode.h, nothing else): free functions with `at::Tensor` out-params first and an optional raw
// `hip_stream` (RegisterBaseBindings.hpp:15-34), and op classes of the reference's shape
//     Op(const AttentionConfigs&);  ParamsPtr prepare(PyAttentionInputs);  Tensor forward(qkv, kv_cache, params)
// (bindings/rocm/FusedRopeKVCacheOp.h:51-61, registration FusedRopeKVCacheOp.cc:648-689), params objects with
// update_kv_cache_offset / prepare_in_place for graph replay.  Everything enqueues on the current torch HIP stream
// (Torch_ext.h:16 GET_CURRENT_STREAM), never synchronises inside forward, and converts a negative C-ABI status into
// TORCH_CHECK -> RuntimeError (Torch_ext.h:48-66).
//
// Built stand-alone as the python module `mi355_compute_ops` (submodule `rtp_llm_ops`), with field-compatible mirrors of
// the three OpDefs.h structs the decode path reads; inside the reference tree the same registerPyModuleOps body is linked
// instead of RegisterRocmOps.cc and the mirrors are replaced by the reference's own types.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include <cmath>
#include <memory>
#include <optional>

#include "../../../include/mi355_decode.h"

// Inside (or against) the reference tree -- -DMI355_REFERENCE_TREE -I<reference root> -- the data contract comes from the
// reference's own headers (both only need torch): AttentionConfigs + RopeConfig, and the ParamsBase the attention ops'
// params objects derive from.  tests/test_reference_plugin.py compiles this file that way when the tree is present.
#ifdef MI355_REFERENCE_TREE
#include "rtp_llm/cpp/model_utils/AttentionConfig.h"
#include "rtp_llm/models_py/bindings/ParamsBase.h"
#endif

namespace py = pybind11;

namespace mi355 {

// ---- mirrors of the data contract (same field names; decode fields only) -------------------------------------------
struct AttentionConfigs {            // rtp_llm/cpp/model_utils/AttentionConfig.h:24-85 (+ RopeConfig.h:7-42: dim, base)
    int64_t head_num = 0, kv_head_num = 0, size_per_head = 0, tokens_per_block = 16, max_seq_len = 8192;
    int64_t rope_dim = 0;
    double  rope_base = 10000.0, softmax_extra_scale = 1.0;
    // RopeConfig.h:7-40: style (1 Base, 3 DynamicNTK, 4 QwenDynamicNTK, 5 Yarn, 6 Llama3; linear = Base with scale), scale = HF factor, factor1 / factor2 =
    // beta_slow / beta_fast (yarn) or low_freq_factor / high_freq_factor (llama3), rope_max_pos = original max positions
    int64_t rope_style = 1, rope_max_pos = 0;
    double  rope_scale = 1.0, rope_factor1 = 1.0, rope_factor2 = 1.0, rope_extrapolation_factor = 1.0, rope_mscale = 1.0;
    bool    use_int8_kv_cache = false;   // KvCacheDataType value 1, the slot the reference removed (AttentionConfig.h:9-12)
};

#ifdef MI355_REFERENCE_TREE
// the reference's own struct -> the fields this path reads
inline AttentionConfigs from_reference(const rtp_llm::AttentionConfigs& r) {
    AttentionConfigs c;
    c.head_num = (int64_t)r.head_num; c.kv_head_num = (int64_t)r.kv_head_num; c.size_per_head = (int64_t)r.size_per_head;
    c.tokens_per_block = (int64_t)(r.kernel_tokens_per_block ? r.kernel_tokens_per_block : r.tokens_per_block);
    c.max_seq_len = (int64_t)r.max_seq_len;
    c.rope_dim = r.rope_config.dim; c.rope_base = r.rope_config.base; c.softmax_extra_scale = r.softmax_extra_scale;
    c.rope_style = (int64_t)r.rope_config.style; c.rope_max_pos = r.rope_config.max_pos; c.rope_scale = r.rope_config.scale;
    c.rope_factor1 = r.rope_config.factor1; c.rope_factor2 = r.rope_config.factor2;
    c.rope_extrapolation_factor = r.rope_config.extrapolation_factor; c.rope_mscale = r.rope_config.mscale;
    TORCH_CHECK(!r.use_mla && r.is_causal && r.dtype == c10::ScalarType::Half, "mi355 decode ops: MHA / GQA, causal, fp16 only");
    TORCH_CHECK(r.kv_cache_dtype == rtp_llm::KvCacheDataType::BASE, "mi355 decode ops: fp16 or INT8 KV cache (the INT8 slot is keyed "
                "off the cache tensor's dtype, see kv_of)");
    return c;
}
#endif

// fp32 {cos, sin} table [max_seq_len][hd / 2][2] of the configured style, on the host (genBaseCache / genYarnCache,
// RopeCache.cc:16-81; Llama3Rope / YarnRope / LinearScaleRope of rotary_position_embedding.h:355-442 tabulated the same way)
inline torch::Tensor build_rope_table(const AttentionConfigs& c) {
    const int64_t hd = c.size_per_head;
    auto idx  = torch::arange(0, hd, 2).to(torch::kFloat32);
    auto step = 1.0 / torch::pow(torch::tensor((float)c.rope_base), idx / (double)hd);
    double gain = 1.0;
    const double pi = 3.14159265358979323846;
    if (c.rope_style == 1) {                         // Base (+ linear position scale)
        if (c.rope_scale != 1.0) step = step / c.rope_scale;
    } else if (c.rope_style == 6) {                  // Llama3: factor1 = low_freq_factor, factor2 = high_freq_factor
        const double ctx = (double)c.rope_max_pos;
        auto wavelen = 2 * pi / step;
        auto blend   = (ctx / wavelen - c.rope_factor1) / (c.rope_factor2 - c.rope_factor1);
        auto banded  = (1 - blend) * step / c.rope_scale + blend * step;
        step = torch::where(wavelen < ctx / c.rope_factor2, step, torch::where(wavelen > ctx / c.rope_factor1, step / c.rope_scale, banded));
    } else if (c.rope_style == 5) {                  // Yarn: factor1 = beta_slow, factor2 = beta_fast
        auto chan = [&](int rotations) {
            return (double)hd * std::log((double)c.rope_max_pos / (rotations * 2 * pi)) / (2 * std::log((double)(int64_t)c.rope_base));
        };
        double first = std::max(std::floor(chan((int)c.rope_factor2)), 0.0), last = std::min(std::ceil(chan((int)c.rope_factor1)), (double)hd - 1);
        if (first == last) last += 0.001;
        auto keep = (1 - torch::clamp((torch::arange(hd / 2).to(torch::kFloat32) - first) / (last - first), 0, 1)) * c.rope_extrapolation_factor;
        step = (step / c.rope_scale) * (1 - keep) + step * keep;
        gain = c.rope_mscale;
    } else if (c.rope_style == 3 || c.rope_style == 4) {
        // DynamicNTK (3) / QwenDynamicNTK (4), rotary_position_embedding.h:889-902 + :925-951: past the original context the base grows with
        // seq_len, and the DECODE writer passes the cached length -- the new token's position -- as seq_len
        // (fused_rope_kvcache_kernel.cu:1341-1392): one base per table row, angle = p / base_p ^ (2 i / hd) (rope_inv_freq, :324-327)
        auto pos   = torch::arange(c.max_seq_len).to(torch::kFloat32);
        auto power = torch::tensor((float)((double)hd / ((double)hd - 2.0)));
        auto base0 = torch::full({c.max_seq_len}, (float)c.rope_base);
        const double ctx = (double)c.rope_max_pos;
        TORCH_CHECK(ctx > 0, "dynamic-NTK rope: rope_config.max_pos (the original context) must be set");
        torch::Tensor grown;
        if (c.rope_style == 3) {
            auto f = torch::tensor((float)c.rope_scale);
            grown = base0 * torch::pow(f * pos / ctx - (f - 1.0), power);
        } else {
            auto octave = torch::ceil(torch::log(torch::clamp_min(pos, 1.0) / ctx) / std::log(2.0) + 1.0);
            grown = base0 * torch::pow(torch::clamp_min(torch::exp2(octave) - 1.0, 1.0), power);
        }
        auto bases = torch::where(pos > ctx, grown, base0);
        auto angle = pos.unsqueeze(1) / torch::pow(bases.unsqueeze(1), (idx / (double)hd).unsqueeze(0));
        return torch::stack({angle.cos(), angle.sin()}, -1).contiguous();
    } else {
        TORCH_CHECK(false, "rope style ", c.rope_style, ": Base / linear (1), DynamicNTK (3), QwenDynamicNTK (4), Yarn (5) and Llama3 (6) are tabulated by position");
    }
    auto freqs = torch::outer(torch::arange(c.max_seq_len).to(torch::kFloat32), step);
    return torch::stack({freqs.cos() * gain, freqs.sin() * gain}, -1).contiguous();
}

struct LayerKVCache {                // bindings/OpDefs.h:29-51
    torch::Tensor kv_cache_base, kv_scale_base;
    int           seq_size_per_block = 0, layer_id = -1;
};

struct PyAttentionInputs {           // bindings/OpDefs.h:281-327
    bool          is_prefill = false, is_target_verify = false, is_cuda_graph = false;
    torch::Tensor sequence_lengths, input_lengths, kv_cache_kernel_block_id, kv_cache_kernel_block_id_device;
    torch::Tensor sequence_lengths_plus_1_device;
};

inline void* cur_stream(int64_t hip_stream = 0) {
    return hip_stream ? reinterpret_cast<void*>(hip_stream) : static_cast<void*>(c10::hip::getCurrentHIPStream().stream());
}
inline void check(int rc, const char* what) { TORCH_CHECK(rc >= 0, what, ": mi355 error ", rc, ": ", mi355_last_error()); }
inline void need(const torch::Tensor& t, c10::ScalarType dt, const char* name) {
    TORCH_CHECK(t.defined() && t.is_cuda(), name, ": tensor must live on the GPU (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == dt, name, ": expected ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": tensor must be contiguous");
}

// activation tensors: fp16 or bf16 (the reference passes either through the same ops); returns the C-ABI dtype code
inline int act_of(const torch::Tensor& t, const char* name) {
    TORCH_CHECK(t.defined() && t.is_cuda(), name, ": tensor must live on the GPU (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == torch::kFloat16 || t.scalar_type() == torch::kBFloat16, name, ": expected float16 or bfloat16, got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": tensor must be contiguous");
    return t.scalar_type() == torch::kBFloat16 ? MI355_ACT_BF16 : MI355_ACT_F16;
}

// ---- params object shared by the two attention ops (the role of CKAttn, FusedRopeKVCacheOp.cc:648-653) -------------
#ifdef MI355_REFERENCE_TREE
struct AttnParams: public rtp_llm::ParamsBase {   // ParamsBase.h:8-22: the framework refills recycled params through fillParams
    void fillParams(torch::Tensor sequence_lengths, torch::Tensor input_lengths, torch::Tensor kv_cache_block_id_host, int batch_size,
                    int seq_size_per_block, torch::Tensor prefix_lengths = torch::Tensor()) override {
        (void)input_lengths; (void)seq_size_per_block; (void)prefix_lengths;
        TORCH_CHECK(positions.defined() && batch_size <= positions.size(0), "fillParams: batch exceeds the prepared params");
        positions.slice(0, 0, batch_size).copy_(sequence_lengths.slice(0, 0, batch_size).to(torch::kInt32), true);
        torch::add_out(seq_lens, positions, 1);
        if (kv_cache_block_id_host.defined())
            block_table.slice(0, 0, batch_size).copy_(kv_cache_block_id_host.slice(0, 0, batch_size).to(torch::kInt32), true);
    }
#else
struct AttnParams {
#endif
    torch::Tensor positions;     // int32 [B] device: tokens already in the cache (= sequence_lengths)
    torch::Tensor seq_lens;      // int32 [B] device: positions + 1
    torch::Tensor block_table;   // int32 [B, M] device

    void update_kv_cache_offset(const torch::Tensor& kv_cache_block_id_device) {
        need(kv_cache_block_id_device, torch::kInt32, "update_kv_cache_offset.block_ids");
        block_table.copy_(kv_cache_block_id_device, /*non_blocking=*/true);      // address-stable refresh for graph replay
    }
    void prepare_in_place(const PyAttentionInputs& in) {
        positions.copy_(in.sequence_lengths.to(torch::kInt32), true);
        torch::add_out(seq_lens, positions, 1);
        if (in.kv_cache_kernel_block_id_device.defined()) update_kv_cache_offset(in.kv_cache_kernel_block_id_device);
    }
};
using AttnParamsPtr = std::shared_ptr<AttnParams>;

inline AttnParamsPtr make_params(const PyAttentionInputs& in) {
    TORCH_CHECK(!in.is_prefill, "mi355 decode ops: is_prefill must be false (prefill and decode are never mixed, "
                                "PyWrappedModel.cc:232-245)");
    need(in.kv_cache_kernel_block_id_device, torch::kInt32, "attn_inputs.kv_cache_kernel_block_id_device");
    TORCH_CHECK(in.sequence_lengths.defined(), "attn_inputs.sequence_lengths missing");
    auto p = std::make_shared<AttnParams>();
    const auto dev = in.kv_cache_kernel_block_id_device.device();
    p->positions   = in.sequence_lengths.to(dev, torch::kInt32).contiguous();   // host work is allowed in prepare()
    p->seq_lens    = (p->positions + 1).contiguous();
    p->block_table = in.kv_cache_kernel_block_id_device.dim() == 2 ? in.kv_cache_kernel_block_id_device
                                                                   : in.kv_cache_kernel_block_id_device.reshape({p->positions.size(0), -1});
    return p;
}

inline mi355_kv_layer_t kv_of(const AttentionConfigs& c, const LayerKVCache& kv, int act = MI355_ACT_F16) {
    TORCH_CHECK(kv.kv_cache_base.defined() && kv.kv_cache_base.is_cuda(), "kv_cache.kv_cache_base must be a GPU tensor");
    const bool int8 = kv.kv_cache_base.scalar_type() == torch::kInt8;
    const bool bf_cache = kv.kv_cache_base.scalar_type() == torch::kBFloat16;
    TORCH_CHECK(int8 || bf_cache || kv.kv_cache_base.scalar_type() == torch::kFloat16, "kv cache dtype must be float16, bfloat16 or int8");
    TORCH_CHECK(int8 || bf_cache == (act == MI355_ACT_BF16), "a 16-bit kv cache has the dtype of the rows it serves");
    TORCH_CHECK(!int8 || kv.kv_scale_base.defined(), "int8 KV cache needs kv_scale_base (fp32 scale plane)");
    mi355_kv_layer_t k;
    k.kv_base    = kv.kv_cache_base.data_ptr();
    k.scale_base = int8 ? kv.kv_scale_base.data_ptr<float>() : nullptr;
    k.kv_dtype   = int8 ? MI355_KV_INT8 : (bf_cache ? MI355_KV_BF16 : MI355_KV_FP16);
    k.page       = kv.seq_size_per_block > 0 ? kv.seq_size_per_block : (int)c.tokens_per_block;
    k.nkv        = (int)c.kv_head_num;
    k.hd         = (int)c.size_per_head;
    k.num_blocks = (int)(kv.kv_cache_base.numel() / (2 * c.kv_head_num * k.page * c.size_per_head));
    k.act_dtype  = act;
    return k;
}

// ---- bias + RoPE + Q-extract + paged KV write (FusedRopeKVCacheDecodeOp*, FusedRopeKVCacheOp.cc:474-646) -----------
class Mi355RopeKVCacheDecodeOp {
public:
    explicit Mi355RopeKVCacheDecodeOp(const AttentionConfigs& c): cfg_(c) {
        TORCH_CHECK(c.head_num > 0 && c.kv_head_num > 0 && (c.size_per_head == 64 || c.size_per_head == 128),
                    "Mi355RopeKVCacheDecodeOp: unsupported head configuration");
        TORCH_CHECK(c.rope_dim == 0 || c.rope_dim == c.size_per_head, "rope_dim must equal size_per_head (full-width rotation)");
        cos_sin_host_ = build_rope_table(c);
    }
#ifdef MI355_REFERENCE_TREE
    explicit Mi355RopeKVCacheDecodeOp(const rtp_llm::AttentionConfigs& r): Mi355RopeKVCacheDecodeOp(from_reference(r)) {}
#endif
    AttnParamsPtr prepare(const PyAttentionInputs& in) { return make_params(in); }

    torch::Tensor forward(const torch::Tensor& qkv, std::optional<LayerKVCache> kv_cache, const AttnParamsPtr& params) {
        const int act = act_of(qkv, "qkv");
        TORCH_CHECK(kv_cache.has_value(), "Mi355RopeKVCacheDecodeOp.forward needs a LayerKVCache");
        TORCH_CHECK(params, "params is null: call prepare() first");
        const int T = (int)qkv.size(0);
        if (!cos_sin_.defined() || cos_sin_.device() != qkv.device()) cos_sin_ = cos_sin_host_.to(qkv.device());
        if (!oob_.defined() || oob_.device() != qkv.device()) oob_ = torch::zeros({1}, qkv.options().dtype(torch::kInt32));
        auto q = torch::empty({T, cfg_.head_num, cfg_.size_per_head}, qkv.options());   // FusedRopeKVCacheOp.cc:538-539
        const mi355_kv_layer_t kv = kv_of(cfg_, *kv_cache, act);
        check(mi355_rope_kv_write(qkv.data_ptr(), nullptr, 0, (int)qkv.size(1), nullptr, cos_sin_.data_ptr<float>(),
                                  (int)cfg_.size_per_head, (int)cos_sin_.size(0), params->positions.data_ptr<int32_t>(),
                                  params->block_table.data_ptr<int32_t>(), (int)params->block_table.size(1), T,
                                  (int)cfg_.head_num, &kv, q.data_ptr(), oob_.data_ptr<int32_t>(), cur_stream()),
              "mi355_rope_kv_write");
        return q;
    }
    int64_t oob_count() const { return oob_.defined() ? oob_.item<int32_t>() : 0; }   // synchronises; diagnostics only

private:
    AttentionConfigs cfg_;
    torch::Tensor    cos_sin_host_, cos_sin_, oob_;
};

// ---- paged decode attention (AiterDecodeAttnOp* / paged_attention_atrex, aiter.py:1340-1561, atrexPA.cc:444-496) ---
class Mi355PagedAttnDecodeOp {
public:
    explicit Mi355PagedAttnDecodeOp(const AttentionConfigs& c): cfg_(c) {
        TORCH_CHECK(c.head_num % c.kv_head_num == 0 && c.head_num / c.kv_head_num <= 16, "GQA group must be <= 16");
    }
#ifdef MI355_REFERENCE_TREE
    explicit Mi355PagedAttnDecodeOp(const rtp_llm::AttentionConfigs& r): Mi355PagedAttnDecodeOp(from_reference(r)) {}
#endif
    AttnParamsPtr prepare(const PyAttentionInputs& in) { return make_params(in); }

    torch::Tensor forward(const torch::Tensor& q, std::optional<LayerKVCache> kv_cache, const AttnParamsPtr& params) {
        const int act = act_of(q, "q");
        TORCH_CHECK(kv_cache.has_value() && params, "Mi355PagedAttnDecodeOp.forward needs a LayerKVCache and params");
        const int B = (int)q.size(0);
        const mi355_kv_layer_t kv = kv_of(cfg_, *kv_cache, act);
        const size_t need_ws = mi355_paged_attn_workspace_bytes(B, (int)cfg_.head_num, (int)cfg_.size_per_head, (int)cfg_.max_seq_len);
        if (!ws_.defined() || ws_.device() != q.device() || (size_t)ws_.numel() < need_ws)
            ws_ = torch::empty({(int64_t)std::max<size_t>(need_ws, 1)}, q.options().dtype(torch::kUInt8));   // address-stable after warm-up
        auto out = torch::empty({B, cfg_.head_num * cfg_.size_per_head}, q.options());
        const float scale = (float)(cfg_.softmax_extra_scale / std::sqrt((double)cfg_.size_per_head));       // PagedAttn.cc:114
        check(mi355_paged_decode_attn(q.data_ptr(), &kv, params->block_table.data_ptr<int32_t>(), (int)params->block_table.size(1),
                                      params->seq_lens.data_ptr<int32_t>(), B, (int)cfg_.head_num, scale, (int)cfg_.max_seq_len,
                                      out.data_ptr(), ws_.data_ptr(), (size_t)ws_.numel(), cur_stream()),
              "mi355_paged_decode_attn");
        return out;
    }

private:
    AttentionConfigs cfg_;
    torch::Tensor    ws_;
};

// ---- weight-only linear: the native object behind the Python strategy classes (DenseWeights{kernel,scales,zeros},
// cpp/models/models_weight/Weights.h:23-29, after the load-time repack of rtp_llm_amd.quant) --------------------------
class Mi355WeightOnlyLinear {
public:
    Mi355WeightOnlyLinear(torch::Tensor qweight, std::optional<torch::Tensor> meta, int64_t wbits, int64_t K, int64_t N,
                          int64_t K_pad, int64_t N_pad, int64_t group_size, std::optional<torch::Tensor> bias):
        qweight_(std::move(qweight)), meta_(meta.value_or(torch::Tensor())), bias_(bias.value_or(torch::Tensor())) {
        TORCH_CHECK(qweight_.is_cuda() && qweight_.is_contiguous(), "qweight must be a contiguous GPU tensor");
        w_.qweight = qweight_.data_ptr();
        w_.meta    = meta_.defined() ? meta_.data_ptr() : nullptr;
        w_.wbits = (int)wbits; w_.K = (int)K; w_.N = (int)N; w_.K_pad = (int)K_pad; w_.N_pad = (int)N_pad; w_.group_size = (int)group_size; w_.act_dtype = MI355_ACT_F16;
        if (bias_.defined()) act_of(bias_, "bias");
    }
    torch::Tensor forward(const torch::Tensor& x, int64_t epilogue) {
        mi355_weight_t w_ = this->w_;                        // per call: the activation dtype of THIS x
        w_.act_dtype = act_of(x, "x");
        TORCH_CHECK(!bias_.defined() || bias_.scalar_type() == x.scalar_type(), "linear: bias dtype must match x");
        TORCH_CHECK(w_.wbits != 16 || qweight_.scalar_type() == x.scalar_type(), "linear: a 16-bit weight image has the dtype it was packed with");
        TORCH_CHECK(x.size(-1) == w_.K, "linear: x last dim ", x.size(-1), " != K ", w_.K);
        const int M = (int)(x.numel() / w_.K);
        auto shape = x.sizes().vec();
        shape.back() = (epilogue & MI355_EPI_SILU_MUL) ? w_.N / 2 : w_.N;
        auto y = torch::empty(shape, x.options().dtype((epilogue & MI355_EPI_OUT_F32) ? torch::kFloat32 : x.scalar_type()));
        const size_t need_ws = mi355_linear_workspace_bytes(M, &w_);
        if (!ws_.defined() || ws_.device() != x.device() || (size_t)ws_.numel() < need_ws)
            ws_ = torch::empty({(int64_t)std::max<size_t>(need_ws, 1 << 20)}, x.options().dtype(torch::kUInt8));
        check(mi355_linear_forward(x.data_ptr(), M, &w_, bias_.defined() ? bias_.data_ptr() : nullptr, y.data_ptr(), (int)epilogue,
                                   ws_.data_ptr(), (size_t)ws_.numel(), cur_stream()),
              "mi355_linear_forward");
        return y;
    }

private:
    torch::Tensor  qweight_, meta_, bias_, ws_;
    mi355_weight_t w_;
};

// ---- free functions, reference argument convention: out-params first, optional raw stream --------------------------
void rmsnorm(torch::Tensor& output, const torch::Tensor& input, const torch::Tensor& weight, double eps, int64_t hip_stream) {
    const int act = act_of(input, "input");
    need(weight, input.scalar_type(), "weight"); need(output, input.scalar_type(), "output");
    const int H = (int)input.size(-1);
    check(mi355_rmsnorm_dt(input.data_ptr(), weight.data_ptr(), (float)eps, (int)(input.numel() / H), H, output.data_ptr(), act, cur_stream(hip_stream)),
          "mi355_rmsnorm");
}
// (normed, residual_out) <- (input (+bias) + residual): RMSResNorm (modules/base/rocm/norm.py:59-77)
void fused_add_rmsnorm(torch::Tensor& output, torch::Tensor& residual_out, const torch::Tensor& input, const torch::Tensor& residual,
                       const torch::Tensor& weight, double eps, std::optional<torch::Tensor> bias, int64_t hip_stream) {
    const int act = act_of(input, "input");
    need(residual, input.scalar_type(), "residual"); need(weight, input.scalar_type(), "weight");
    need(output, input.scalar_type(), "output"); need(residual_out, input.scalar_type(), "residual_out");
    if (bias) need(*bias, input.scalar_type(), "bias");
    const int H = (int)input.size(-1);
    check(mi355_add_rmsnorm_dt(input.data_ptr(), nullptr, 0, 0, bias ? bias->data_ptr() : nullptr, residual.data_ptr(), residual_out.data_ptr(),
                               weight.data_ptr(), (float)eps, (int)(input.numel() / H), H, output.data_ptr(), act, cur_stream(hip_stream)),
          "mi355_add_rmsnorm");
}
void silu_and_mul(torch::Tensor& output, const torch::Tensor& gate_up, int64_t hip_stream) {   // aiter.silu_and_mul(out, x)
    const int act = act_of(gate_up, "gate_up");
    need(output, gate_up.scalar_type(), "output");
    const int I = (int)gate_up.size(-1) / 2;
    check(mi355_silu_mul_dt(gate_up.data_ptr(), (int)(gate_up.numel() / (2 * I)), I, output.data_ptr(), act, cur_stream(hip_stream)), "mi355_silu_mul");
}
void embedding(torch::Tensor& output, const torch::Tensor& input, const torch::Tensor& weight) {   // RegisterBaseBindings.hpp:36-43
    need(input, torch::kInt32, "input"); act_of(weight, "weight"); need(output, weight.scalar_type(), "output");
    check(mi355_embedding(input.data_ptr<int32_t>(), (int)input.numel(), weight.data_ptr(), (int)weight.size(1), (int)weight.size(0),
                          output.data_ptr(), cur_stream()),
          "mi355_embedding");
}
torch::Tensor greedy_argmax(const torch::Tensor& logits) {   // the top_k == 1 fast path of sampleGreedy (CudaSampleOp.cc:687-700)
    need(logits, torch::kFloat32, "logits");
    const int B = (int)logits.size(0), V = (int)logits.size(1);
    auto ids = torch::empty({B}, logits.options().dtype(torch::kInt32));
    auto ws  = torch::empty({(int64_t)B * 64 * 8}, logits.options().dtype(torch::kUInt8));
    check(mi355_argmax(logits.data_ptr<float>(), B, V, V, ids.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(), cur_stream()), "mi355_argmax");
    return ids;
}

}  // namespace mi355

namespace rtp_llm {

// The hook the reference links per build flavour (RegisterOps.h:9).
void registerPyModuleOps(py::module& m) {
    using namespace mi355;
    using AttentionConfigs = mi355::AttentionConfigs;   // (rtp_llm::AttentionConfigs is in scope too when built against the tree)
#ifdef MI355_REFERENCE_TREE
    // the tree registers its own AttentionConfigs / PyAttentionInputs bindings (RegisterBaseBindings.hpp): the mirror keeps a
    // distinct Python name and the ops also construct from the tree's struct
    py::class_<AttentionConfigs>(m, "Mi355AttentionConfigs")
#else
    py::class_<AttentionConfigs>(m, "AttentionConfigs")
#endif
        .def(py::init<>())
        .def_readwrite("head_num", &AttentionConfigs::head_num)
        .def_readwrite("kv_head_num", &AttentionConfigs::kv_head_num)
        .def_readwrite("size_per_head", &AttentionConfigs::size_per_head)
        .def_readwrite("tokens_per_block", &AttentionConfigs::tokens_per_block)
        .def_readwrite("max_seq_len", &AttentionConfigs::max_seq_len)
        .def_readwrite("rope_dim", &AttentionConfigs::rope_dim)
        .def_readwrite("rope_base", &AttentionConfigs::rope_base)
        .def_readwrite("softmax_extra_scale", &AttentionConfigs::softmax_extra_scale)
        .def_readwrite("rope_style", &AttentionConfigs::rope_style)
        .def_readwrite("rope_max_pos", &AttentionConfigs::rope_max_pos)
        .def_readwrite("rope_scale", &AttentionConfigs::rope_scale)
        .def_readwrite("rope_factor1", &AttentionConfigs::rope_factor1)
        .def_readwrite("rope_factor2", &AttentionConfigs::rope_factor2)
        .def_readwrite("rope_extrapolation_factor", &AttentionConfigs::rope_extrapolation_factor)
        .def_readwrite("rope_mscale", &AttentionConfigs::rope_mscale)
        .def_readwrite("use_int8_kv_cache", &AttentionConfigs::use_int8_kv_cache);
    py::class_<LayerKVCache>(m, "LayerKVCache")
        .def(py::init<>())
        .def_readwrite("kv_cache_base", &LayerKVCache::kv_cache_base)
        .def_readwrite("kv_scale_base", &LayerKVCache::kv_scale_base)
        .def_readwrite("seq_size_per_block", &LayerKVCache::seq_size_per_block)
        .def_readwrite("layer_id", &LayerKVCache::layer_id);
    py::class_<PyAttentionInputs>(m, "PyAttentionInputs")
        .def(py::init<>())
        .def_readwrite("is_prefill", &PyAttentionInputs::is_prefill)
        .def_readwrite("is_target_verify", &PyAttentionInputs::is_target_verify)
        .def_readwrite("is_cuda_graph", &PyAttentionInputs::is_cuda_graph)
        .def_readwrite("sequence_lengths", &PyAttentionInputs::sequence_lengths)
        .def_readwrite("input_lengths", &PyAttentionInputs::input_lengths)
        .def_readwrite("kv_cache_kernel_block_id", &PyAttentionInputs::kv_cache_kernel_block_id)
        .def_readwrite("kv_cache_kernel_block_id_device", &PyAttentionInputs::kv_cache_kernel_block_id_device)
        .def_readwrite("sequence_lengths_plus_1_device", &PyAttentionInputs::sequence_lengths_plus_1_device);
    py::class_<AttnParams, AttnParamsPtr>(m, "Mi355AttnParams")
        .def(py::init<>())
        .def("update_kv_cache_offset", &AttnParams::update_kv_cache_offset, py::arg("kv_cache_block_id_device"))
        .def("prepare_in_place", &AttnParams::prepare_in_place, py::arg("attn_inputs"))
        .def_readonly("positions", &AttnParams::positions)
        .def_readonly("seq_lens", &AttnParams::seq_lens)
        .def_readonly("block_table", &AttnParams::block_table);
    py::class_<Mi355RopeKVCacheDecodeOp>(m, "Mi355RopeKVCacheDecodeOp")
        .def(py::init<const AttentionConfigs&>(), py::arg("attn_configs"))
#ifdef MI355_REFERENCE_TREE
        .def(py::init<const rtp_llm::AttentionConfigs&>(), py::arg("attn_configs"))
#endif
        .def("prepare", &Mi355RopeKVCacheDecodeOp::prepare, py::arg("attn_inputs"))
        .def("forward", &Mi355RopeKVCacheDecodeOp::forward, py::arg("qkv"), py::arg("kv_cache"), py::arg("params"))
        .def("oob_count", &Mi355RopeKVCacheDecodeOp::oob_count);
    py::class_<Mi355PagedAttnDecodeOp>(m, "Mi355PagedAttnDecodeOp")
        .def(py::init<const AttentionConfigs&>(), py::arg("attn_configs"))
#ifdef MI355_REFERENCE_TREE
        .def(py::init<const rtp_llm::AttentionConfigs&>(), py::arg("attn_configs"))
#endif
        .def("prepare", &Mi355PagedAttnDecodeOp::prepare, py::arg("attn_inputs"))
        .def("forward", &Mi355PagedAttnDecodeOp::forward, py::arg("q"), py::arg("kv_cache"), py::arg("params"));
    py::class_<Mi355WeightOnlyLinear>(m, "Mi355WeightOnlyLinear")
        .def(py::init<torch::Tensor, std::optional<torch::Tensor>, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                      std::optional<torch::Tensor>>(),
             py::arg("qweight"), py::arg("meta"), py::arg("wbits"), py::arg("K"), py::arg("N"), py::arg("K_pad"), py::arg("N_pad"),
             py::arg("group_size"), py::arg("bias") = py::none())
        .def("forward", &Mi355WeightOnlyLinear::forward, py::arg("x"), py::arg("epilogue") = 0);
    m.def("rmsnorm", &rmsnorm, "RMSNorm", py::arg("output"), py::arg("input"), py::arg("weight"), py::arg("eps"), py::arg("hip_stream") = 0);
    m.def("fused_add_rmsnorm", &fused_add_rmsnorm, "residual add + RMSNorm", py::arg("output"), py::arg("residual_out"), py::arg("input"),
          py::arg("residual"), py::arg("weight"), py::arg("eps"), py::arg("bias") = py::none(), py::arg("hip_stream") = 0);
    m.def("silu_and_mul", &silu_and_mul, "SiLU-gate", py::arg("output"), py::arg("gate_up"), py::arg("hip_stream") = 0);
    m.def("embedding", &embedding, "Embedding lookup kernel", py::arg("output"), py::arg("input"), py::arg("weight"));
    m.def("greedy_argmax", &greedy_argmax, "argmax over fp32 logits, lowest index on ties", py::arg("logits"));
    m.def("rope_table", &build_rope_table, "fp32 {cos, sin} table [max_seq_len][hd / 2][2] of the configured RoPE style", py::arg("attn_configs"));
    m.def("abi_version", []() { return mi355_abi_version(); });
}

}  // namespace rtp_llm

PYBIND11_MODULE(mi355_compute_ops, m) {
    m.doc() = "MI355X-native decode ops behind the reference's registerPyModuleOps hook";
    auto ops = m.def_submodule("rtp_llm_ops", "same place the reference exposes its ops: <module>.rtp_llm_ops");
    rtp_llm::registerPyModuleOps(ops);
}
