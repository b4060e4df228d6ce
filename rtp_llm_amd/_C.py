"""ctypes binding of libmi355_decode.so (the C-ABI of include/mi355_decode.h).

The library is the product: if it is missing, importing ``lib()`` raises — there is
no CPU or PyTorch fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_TUNING_LIB=1 (set by tools/ and bench.py --debug-set only) selects the tuning build, which adds process-global
# experiment switches (mi355_debug_set & co); the product library has none.
TUNING = os.environ.get("MI355_TUNING_LIB") == "1"
LIB_PATH = os.path.join(_HERE, "lib", "libmi355_decode_tuning.so" if TUNING else "libmi355_decode.so")

OK, ERR_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_WORKSPACE = 0, -1, -2, -3, -4
W4, W8, W16 = 4, 8, 16
KV_FP16, KV_INT8, KV_BF16 = 0, 1, 2
ACT_F16, ACT_BF16 = 0, 1
EPI_NONE, EPI_SILU_MUL, EPI_OUT_F32, EPI_OUT_IMAGE = 0, 1, 2, 4
PF_QKV, PF_O, PF_GATE_UP, PF_QKV_LATE, PF_O_LATE, PF_TP_COMM, PF_TP_INLAUNCH, PF_QKV_IN_FOLD = 1, 2, 4, 16, 32, 64, 128, 256   # mi355_decoder_set_weight_prefetch mask bits
HINT_STAGED, HINT_NO_PERSISTENT = 0x100, 0x200
ABI_VERSION = 3
KC_NAMES = ["gemm_quant", "gemm_lmhead", "attn", "rope_kv", "norm", "other", "comm"]

vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t


class Weight(C.Structure):  # mi355_weight_t
    _fields_ = [("qweight", vp), ("meta", vp), ("wbits", i32), ("K", i32), ("N", i32),
                ("K_pad", i32), ("N_pad", i32), ("group_size", i32), ("act_dtype", i32)]


class KVLayer(C.Structure):  # mi355_kv_layer_t
    _fields_ = [("kv_base", vp), ("scale_base", vp), ("kv_dtype", i32), ("page", i32),
                ("nkv", i32), ("hd", i32), ("num_blocks", i32), ("act_dtype", i32)]


class FusedNorm(C.Structure):  # mi355_fused_norm_t
    _fields_ = [("tile_sumsq", vp), ("tiles", i32), ("ld", i32), ("weight", vp), ("eps", f32)]


class DeferredNorm(C.Structure):  # mi355_deferred_norm_t
    _fields_ = [("tile_sumsq", vp), ("tiles", i32), ("ld", i32), ("eps", f32), ("unscale", f32)]


class ModelConfig(C.Structure):  # mi355_model_config_t
    _fields_ = [(n, i32) for n in ("num_layers", "hidden", "nh", "nkv", "hd", "inter", "vocab", "rope_dim", "max_pos")] + \
               [("rms_eps", f32)] + \
               [(n, i32) for n in ("kv_dtype", "page", "num_blocks", "max_batch", "max_blocks_per_seq", "max_seq_len", "tp_size", "act_dtype")]


class LayerWeights(C.Structure):  # mi355_layer_weights_t
    _fields_ = [("qkv", Weight), ("o", Weight), ("gate_up", Weight), ("down", Weight),
                ("qkv_bias", vp), ("input_norm", vp), ("post_norm", vp), ("kv_base", vp), ("kv_scale_base", vp)]


class ModelWeights(C.Structure):  # mi355_model_weights_t
    _fields_ = [("embedding", vp), ("vocab_full", i32), ("final_norm", vp), ("lm_head", Weight), ("cos_sin", vp)]


class StepBuffers(C.Structure):  # mi355_step_buffers_t
    _fields_ = [("token_ids", vp), ("positions", vp), ("block_table", vp), ("logits", vp), ("hidden", vp),
                ("ar_buf", vp), ("workspace", vp), ("workspace_bytes", sz)]


ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Collective(C.Structure):
    """mi355_collective_t: an external transport for the TP points of the step (RCCL fallback)."""
    _fields_ = [("ctx", C.c_void_p), ("all_reduce_f16", ALL_REDUCE_FN), ("all_reduce_bf16", ALL_REDUCE_FN), ("all_gather", ALL_GATHER_FN),
                ("rank", C.c_int32), ("world", C.c_int32)]


# symbol -> (restype, argtypes); every symbol include/mi355_decode.h declares
SIGNATURES = {
    "mi355_abi_version": (i32, []),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_linear_workspace_bytes": (sz, [i32, C.POINTER(Weight)]),
    "mi355_linear_forward": (i32, [vp, i32, C.POINTER(Weight), vp, vp, i32, vp, sz, vp]),
    "mi355_linear_partial": (i32, [vp, i32, C.POINTER(Weight), vp, i32, vp]),
    "mi355_rmsnorm": (i32, [vp, vp, f32, i32, i32, vp, vp]),
    "mi355_add_rmsnorm": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, f32, i32, i32, vp, vp]),
    "mi355_silu_mul": (i32, [vp, i32, i32, vp, vp]),
    "mi355_rmsnorm_dt": (i32, [vp, vp, f32, i32, i32, vp, i32, vp]),
    "mi355_add_rmsnorm_dt": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, f32, i32, i32, vp, i32, vp]),
    "mi355_silu_mul_dt": (i32, [vp, i32, i32, vp, i32, vp]),
    "mi355_embedding": (i32, [vp, i32, vp, i32, i32, vp, vp]),
    "mi355_rope_kv_write": (i32, [vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, i32, i32, i32, C.POINTER(KVLayer), vp, vp, vp]),
    "mi355_rope_kv_write_rows": (i32, [vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, C.POINTER(KVLayer), vp, vp, vp]),
    "mi355_linear_residual": (i32, [vp, i32, C.POINTER(Weight), vp, vp, vp, vp, i32, vp]),
    "mi355_norm_linear": (i32, [vp, i32, C.POINTER(FusedNorm), C.POINTER(Weight), vp, vp, i32, vp]),
    "mi355_act_image_bytes": (sz, [i32, i32]),
    "mi355_act_image_pack": (i32, [vp, i32, i32, vp, i32, i32, vp]),
    "mi355_add_rmsnorm_img": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, f32, i32, i32, vp, i32, vp]),
    "mi355_paged_attn_rows_img": (i32, [vp, C.POINTER(KVLayer), vp, i32, vp, i32, i32, i32, f32, i32, vp, vp, sz, vp]),
    "mi355_linear_residual_img": (i32, [vp, i32, C.POINTER(Weight), vp, vp, vp, vp, i32, vp]),
    "mi355_linear_publish_img": (i32, [vp, i32, C.POINTER(Weight), vp, vp, vp]),
    "mi355_linear_residual_prenorm_img": (i32, [vp, i32, C.POINTER(Weight), vp, vp, vp, vp, i32, vp, vp, i32, vp]),
    "mi355_linear_deferred_norm_img": (i32, [vp, i32, C.POINTER(DeferredNorm), C.POINTER(Weight), vp, vp, i32, vp]),
    "mi355_linear_direct_img": (i32, [vp, i32, C.POINTER(Weight), vp, vp, i32, vp]),
    "mi355_linear_partial_img": (i32, [vp, i32, C.POINTER(Weight), vp, i32, vp]),
    "mi355_qkv_rope_kv_write_img": (i32, [vp, i32, C.POINTER(Weight), vp, vp, i32, i32, vp, vp, i32, i32, i32, C.POINTER(KVLayer), vp, vp, vp]),
    "mi355_qkv_rope_kv_write": (i32, [vp, i32, C.POINTER(Weight), vp, C.POINTER(FusedNorm), vp, i32, i32, vp, vp, i32, i32, i32, C.POINTER(KVLayer), vp, vp, vp]),
    "mi355_paged_attn_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "mi355_paged_decode_attn": (i32, [vp, C.POINTER(KVLayer), vp, i32, vp, i32, i32, f32, i32, vp, vp, sz, vp]),
    "mi355_paged_attn_rows": (i32, [vp, C.POINTER(KVLayer), vp, i32, vp, i32, i32, i32, f32, i32, vp, vp, sz, vp]),
    "mi355_argmax": (i32, [vp, i32, i32, i32, vp, vp, sz, vp]),
    "mi355_softmax_rows": (i32, [vp, i32, i32, i32, f32, vp, vp]),
    "mi355_sample_rows": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "mi355_apply_penalties": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "mi355_ban_repeat_ngram": (i32, [vp, i32, i32, i32, vp, i32, vp, vp, vp]),
    "mi355_top_k_top_p_sample": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    "mi355_rejection_sample": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mi355_allreduce_handle_bytes": (sz, []),
    "mi355_allreduce_create": (vp, [i32, i32, sz, vp]),
    "mi355_allreduce_open": (i32, [vp, vp]),
    "mi355_allreduce_destroy": (None, [vp]),
    "mi355_allreduce_status": (i32, [vp, vp]),
    "mi355_allreduce_set_spin_timeout_ms": (i32, [vp, i32]),
    "mi355_allreduce_set_full_fences": (i32, [vp, i32]),
    "mi355_allreduce_set_protocol": (i32, [vp, i32]),
    "mi355_allreduce_clear_status": (i32, [vp, vp]),
    "mi355_allreduce_sum": (i32, [vp, vp, vp, i32, i32, vp]),
    "mi355_allreduce_fused": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, f32, i32, i32, vp, vp]),
    "mi355_allreduce_fused_dt": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, f32, i32, i32, vp, i32, vp]),
    "mi355_allreduce_fused_img_dt": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, f32, i32, i32, vp, i32, vp]),
    "mi355_allreduce_fused_published_dt": (i32, [vp, vp, vp, vp, f32, i32, i32, vp, i32, i32, vp]),
    "mi355_allreduce_sum_dt": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "mi355_allreduce_argmax": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "mi355_decoder_attach_allreduce": (i32, [vp, vp, i32]),
    "mi355_decoder_set_embedding_split": (i32, [vp, i32]),
    "mi355_decoder_attach_collective": (i32, [vp, C.POINTER(Collective), i32]),
    "mi355_rccl_unique_id_bytes": (sz, []),
    "mi355_rccl_unique_id": (i32, [C.c_char_p, vp]),
    "mi355_rccl_open": (vp, [C.c_char_p, vp, i32, i32]),
    "mi355_rccl_collective": (i32, [vp, C.POINTER(Collective)]),
    "mi355_rccl_close": (None, [vp]),
    "mi355_decoder_set_weight_prefetch": (i32, [vp, i32]),
    "mi355_allreduce_set_prefetch": (i32, [vp, vp, sz]),
    "mi355_allgather_hidden": (i32, [vp, vp, vp, i32, i32, vp]),
    "mi355_decoder_workspace_bytes": (sz, [C.POINTER(ModelConfig)]),
    "mi355_decoder_create": (vp, [C.POINTER(ModelConfig), C.POINTER(LayerWeights), C.POINTER(ModelWeights), C.POINTER(StepBuffers)]),
    "mi355_decoder_destroy": (None, [vp]),
    "mi355_decoder_begin": (i32, [vp, i32, vp]),
    "mi355_decoder_begin_rows": (i32, [vp, i32, i32, vp]),
    "mi355_decoder_layer_attn": (i32, [vp, i32, vp]),
    "mi355_decoder_layer_mlp": (i32, [vp, i32, vp]),
    "mi355_decoder_finish": (i32, [vp, i32, vp]),
    "mi355_decoder_step": (i32, [vp, i32, vp]),
    "mi355_decoder_prefill_workspace_bytes": (sz, [vp, i32, i32]),
    "mi355_decoder_prefill": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]),
    "mi355_decoder_set_prefill_rope_table": (i32, [vp, vp]),
    "mi355_decoder_capture": (i32, [vp, i32]),
    "mi355_decoder_replay": (i32, [vp, i32, i32, vp]),
    "mi355_decoder_profile": (i32, [vp, i32, i32, C.POINTER(f32), C.POINTER(i32), vp]),
    "mi355_decoder_oob_count": (C.c_int64, [vp, vp]),
}

_lib = None


class Mi355Error(RuntimeError):
    pass


def lib():
    """Load the HIP library (once).  Raises if it has not been built: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355Error(f"{LIB_PATH} not found - build it with `python -m rtp_llm_amd.build` "
                             "(the HIP extension is required; there is no CPU fallback)")
        # PyTorch-ROCm bundles its own libamdhip64; it must be loaded first so that this library binds to the
        # same HIP runtime instance (two runtimes in one process cannot share device pointers / streams).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        if l.mi355_abi_version() != ABI_VERSION:
            raise Mi355Error("libmi355_decode.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> int:
    """Status -> RuntimeError, the error behaviour of the reference's ops
    (TORCH_CHECK -> RuntimeError, bindings/common/Torch_ext.h:48-66)."""
    if rc < 0:
        msg = lib().mi355_last_error().decode(errors="replace")
        raise Mi355Error(f"{what}: mi355 error {rc}: {msg}")
    return rc
