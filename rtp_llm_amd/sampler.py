"""The sampler step behind lm_head: host mirror of `sampleGreedy` (rtp_llm/models_py/bindings/core/CudaSampleOp.cc:619-800) over the
MI355 kernels -- same parameter record (bindings/core/OpData.h:260-286, the fields this path reads), same order of operations
and the same host-side shortcuts:

  1. temperature, only when some row has T != 1                          (:633-645)
  2. repetition / presence / frequency penalties, only when some row is off-default   (:648-685)
  2b. no_repeat_ngram_size, only when some decoder row has one                          (:242-283)
  3. every top_k == 1 and no probabilities wanted: arg-max of the logits (:688-700)
  4. softmax in place, top_p == 0 reads as 1                             (:703-736)
  5. every top_k == 1: arg-max of the probabilities; otherwise top-k / top-p filter, renormalise, draw   (:739-786)
  6. the drawn ids go into the last column of token_ids                 (:797-799)

Randomness is explicit: where the reference consumes `generator` inside torch.multinomial, the caller passes one uniform in
[0, 1) per row (`uniform`), so a run is reproducible and checkable against the oracle.  `cum_log_probs` / `output_log_probs`
are not on this path (the reference's ROCm branch does not update them per token either)."""
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops


@dataclass
class GreedyParams:
    logits: torch.Tensor              # [batch_size, vocab_size] fp32, GPU, modified in place (penalties, then probabilities)
    input_lengths: torch.Tensor       # [batch_size] int32
    sequence_lengths: torch.Tensor    # [decoder_batch_size] int32
    token_ids: torch.Tensor           # [batch_size, step + 1] int32; column `step` receives the new token
    step: int
    top_k: torch.Tensor               # [batch_size] int32
    top_p: torch.Tensor               # [batch_size] fp32
    temperature: torch.Tensor         # [batch_size] fp32
    repetition_penalty: Optional[torch.Tensor] = None
    presence_penalty: Optional[torch.Tensor] = None
    frequency_penalty: Optional[torch.Tensor] = None
    no_repeat_ngram_size: Optional[torch.Tensor] = None   # [decoder_batch_size] int32, 0 = off
    output_all_probs: Optional[torch.Tensor] = None     # [batch_size, vocab_size] fp32 GPU: receives the probabilities
    return_original_all_probs: bool = False
    uniform: Optional[torch.Tensor] = None              # [batch_size] fp32 in [0, 1): the draw (see the module docstring)


def _host(t: torch.Tensor, dtype) -> torch.Tensor:
    return t.detach().to(device="cpu", dtype=dtype)


def sample_greedy(params: GreedyParams) -> torch.Tensor:
    """-> the new token ids [batch_size] int32 on the GPU (also written to params.token_ids[:, step])."""
    logits = params.logits
    B, V = logits.shape
    dev = logits.device
    step = int(params.step)
    if params.token_ids.shape != (B, step + 1):
        raise ValueError("sample_greedy: token_ids must be [batch_size, step + 1]")
    transposed = params.token_ids.to(device=dev, dtype=torch.int32).t().contiguous()       # [step + 1, batch_size]

    temperature = _host(params.temperature, torch.float32)
    if bool((temperature != 1.0).any()):
        ops.apply_penalties(logits, temperature=temperature)

    if params.repetition_penalty is not None:
        if params.presence_penalty is None or params.frequency_penalty is None:
            raise ValueError("sample_greedy: repetition_penalty comes with presence_penalty and frequency_penalty")
        rep, pres, freq = (_host(t, torch.float32) for t in (params.repetition_penalty, params.presence_penalty, params.frequency_penalty))
        if bool((rep != 1.0).any()) or bool((pres != 0.0).any()) or bool((freq != 0.0).any()):
            lengths = _host(params.input_lengths, torch.int32).clone()
            nd = params.sequence_lengths.numel()
            if nd > 0:
                lengths[:nd] = _host(params.sequence_lengths, torch.int32)
            ops.apply_penalties(logits, repetition_penalty=rep, presence_penalty=pres, frequency_penalty=freq, output_ids=transposed,
                                input_lengths=lengths, max_input_length=step + 1, step=step + 1)

    if params.no_repeat_ngram_size is not None and params.sequence_lengths.numel() > 0:
        ngram = _host(params.no_repeat_ngram_size, torch.int32)
        nd = params.sequence_lengths.numel()
        if bool((ngram[:nd] != 0).any()):
            # the kernel takes the index of the last valid token and adds one itself (CudaSampleOp.cc:262-265)
            tokens = params.token_ids.to(device=dev, dtype=torch.int32).contiguous()
            ops.ban_repeat_ngram(logits, tokens[:nd], _host(params.sequence_lengths, torch.int32) - 1, ngram[:nd])

    top_k = _host(params.top_k, torch.int32)
    all_greedy = bool((top_k == 1).all())
    if all_greedy and params.output_all_probs is None:
        ids = ops.argmax(logits)
    else:
        probs = ops.softmax_rows(logits)
        logits.copy_(probs)
        top_p = _host(params.top_p, torch.float32).clone()
        top_p[top_p.abs() < 1e-7] = 1.0
        want_renorm = params.output_all_probs is not None and not params.return_original_all_probs
        if all_greedy:
            ids = ops.argmax(probs)
            if want_renorm:   # top_k_renorm_probs with k = 1: the mass of the maximum (ties share it)
                _, renorm = ops.top_k_top_p_sample(probs, top_k, None, torch.zeros(B, device=dev), return_probs=True)
                params.output_all_probs.copy_(renorm)
        else:
            if params.uniform is None:
                raise ValueError("sample_greedy: sampling rows need params.uniform (one value in [0, 1) per row)")
            u = params.uniform.to(device=dev, dtype=torch.float32).contiguous()
            ids, renorm = ops.top_k_top_p_sample(probs, top_k, top_p, u, return_probs=True)
            if want_renorm:
                params.output_all_probs.copy_(renorm)
        if params.return_original_all_probs and params.output_all_probs is not None:
            params.output_all_probs.copy_(probs)
    params.token_ids[:, step] = ids.to(params.token_ids.device)
    return ids
