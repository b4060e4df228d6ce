"""Checkpoint directory -> loader.load_hf_checkpoint -> DecoderEngine (HIP) -> greedy decode, against the oracle run on
the same loaded canonical weights (SURVEY 8f n2 end to end): GPTQ with bf16 auxiliary tensors (as Qwen2 checkpoints ship
them), AWQ, and fp16 + load-time INT8 autoquant."""
import pytest
import torch

from ckpt_util import write_ckpt
from oracle import oracle
from rtp_llm_amd import _C, loader, model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = model.ModelConfig("tiny-ckpt", 2, 512, 8, 2, 64, 1024, 2048, rope_theta=1e6, max_pos=256)


def _dense(c):
    if c.kind == "fp16":
        return c.w.float()
    if c.kind == "int8":
        return oracle.dequant_int8(c.q, c.scales)
    return oracle.dequant_groupwise(c.q, c.z_eff, c.scales, c.group_size)


@pytest.mark.parametrize("kind,quantization,bf16_aux,dtype", [("gptq", None, True, torch.float16), ("awq", None, False, torch.float16),
                                                              ("fp16", "int8", True, torch.float16), ("gptq", None, True, torch.bfloat16),
                                                              ("fp16", None, True, torch.bfloat16)])
def test_checkpoint_to_engine_matches_oracle(tmp_path, kind, quantization, bf16_aux, dtype):
    assert torch.cuda.is_available()
    _C.lib()
    canon = model.synth_model(CFG, "fp16" if kind == "fp16" else "w4", "cpu", seed=31, method="awq" if kind == "awq" else "gptq")
    write_ckpt(str(tmp_path), kind, CFG, canon, bf16_aux=bf16_aux)
    mc, w = loader.load_hf_checkpoint(str(tmp_path), quantization=quantization, dtype=dtype)
    assert w["layers"][0]["qkv"].kind == ("int8" if quantization else ("fp16" if kind == "fp16" else "w4"))
    tol = 1e-2 if dtype == torch.float16 else 3e-2          # bf16: 8 significant bits (tests/test_gpu_bf16.py)
    B, page = 3, 16
    eng = model.DecoderEngine(mc, model.weights_to(w, DEV), kv_int8=False, page=page, num_blocks=32, max_batch=4, max_seq_len=64, device=DEV,
                              dtype=dtype)
    ow = {"embedding": w["embedding"], "final_norm": w["final_norm"], "lm_head": _dense(w["lm_head"]),
          "layers": [{"input_norm": L["input_norm"], "post_norm": L["post_norm"], "qkv_bias": L["qkv_bias"],
                      **{k: _dense(L[k]) for k in ("qkv", "o", "gate_up", "down")}} for L in w["layers"]]}
    odec = oracle.OracleDecoder({**mc.__dict__}, ow)
    okv = oracle.OracleKV(mc.num_layers, B, False)
    bt = torch.arange(B * 4, dtype=torch.int32).reshape(B, 4)
    tok = torch.tensor([3, 700, 1999], dtype=torch.int32)
    eng.set_inputs(tok.tolist(), [0] * B, bt)
    eng.capture(B)
    for step in range(6):
        pos = torch.full((B,), step, dtype=torch.int32)
        _, ref = odec.forward_tokens(tok, pos, okv, list(range(B)))
        eng.replay(B, 1)
        torch.cuda.synchronize()
        got = eng.logits[:B].cpu()
        assert torch.allclose(got, ref, atol=tol, rtol=tol), (step, float((got - ref).abs().max()))
        nxt = oracle.greedy(ref)
        top2 = ref.topk(2, -1).values
        safe = (top2[:, 0] - top2[:, 1]) > tol
        assert torch.equal(eng.token_ids[:B].cpu()[safe], nxt[safe])
        tok = nxt
        eng.token_ids[:B].copy_(tok)
    assert eng.oob_count() == 0
