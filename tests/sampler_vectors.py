"""Known-answer cases of the reference's sampler test, transcribed (inputs and expectations only):
rtp_llm/models_py/bindings/cuda/ops/tests/CudaSamplerTest.cc:518-568 (top_k = 1), :599-657 (top-k), :659-719 (top-p),
:721-780 (top-k + top-p), :905-975 (penalties: the probabilities after repetition / presence / frequency penalty).
Each case: logits [4, 10], token_ids [4, 6] (step = 5), sequence_lengths 5, input_lengths -1, the per-row parameters, and per row
either the exact token or the set of tokens the reference accepts."""
import torch

_LOGITS = [0, 0, 0, 0.1, 0.2, 0.3, 0, 0, 0, 0.01, 0.987, 0.887, 0.99999, 0.1,
           0.2, 0.3, 0, 0, 0.99, 0.989, 0.221, 0, 0, 0.1, 0.2, 0.321, 0, 0.4432,
           0.44, 0.01, 0.221, 0, 0, 0.1, 0.2, 0.321, 0, 0.4432, 0.44, 0.01]
_TOKENS = [100, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0]
_TEMP = [1.0, 10.0, 1.0, 10.0]

CASES = {
    "top_k_1": dict(top_k=[1, 1, 1, 1], top_p=[1.0, 1.0, 1.0, 1.0], allowed=[{5}, {2}, {7}, {7}]),
    "top_k": dict(top_k=[1, 1, 3, 2], top_p=[1.0, 1.0, 1.0, 1.0], allowed=[{5}, {2}, {5, 7, 8}, {7, 8}]),
    "top_p": dict(top_k=[0, 0, 0, 0], top_p=[0.1, 0.1, 0.6, 0.8], allowed=[{5}, {2}, {7, 8, 5, 0, 4, 3}, {7, 8, 5, 0, 4, 3, 9, 1}]),
    "top_k_top_p": dict(top_k=[1, 0, 0, 2], top_p=[0.2, 0.2, 0.6, 0.6], allowed=[{5}, {2, 8}, {7, 8, 5, 0, 4, 3}, {7, 8}]),
}


def case(name):
    c = CASES[name]
    return dict(logits=torch.tensor(_LOGITS, dtype=torch.float32).reshape(4, 10), token_ids=torch.tensor(_TOKENS, dtype=torch.int32).reshape(4, 6),
                step=5, sequence_lengths=torch.full((4,), 5, dtype=torch.int32), input_lengths=torch.full((4,), -1, dtype=torch.int32),
                temperature=torch.tensor(_TEMP), top_k=torch.tensor(c["top_k"], dtype=torch.int32), top_p=torch.tensor(c["top_p"]),
                allowed=c["allowed"])


PENALTY = dict(
    logits=torch.tensor([0.01, 0.88, 0.92, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.01, 0.88, 0.92, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7,
                         0.01, 0.88, 0.92, 0.1, 0.2, 0.3, 0.4, 0.1, 0.1, 0.1, 0.01, 0.88, 0.92, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7]).reshape(4, 10),
    token_ids=torch.tensor([2, 2, 2, 1, 1, 0] * 4, dtype=torch.int32).reshape(4, 6), step=5,
    sequence_lengths=torch.full((4,), 5, dtype=torch.int32), input_lengths=torch.full((4,), -1, dtype=torch.int32),
    temperature=torch.ones(4), top_k=torch.zeros(4, dtype=torch.int32), top_p=torch.ones(4),
    repetition_penalty=torch.tensor([2.4, 1.0, 1.0, 1.2]), presence_penalty=torch.tensor([0, 0.6, 0, 0.3]),
    frequency_penalty=torch.tensor([0, 0, 0.2, 0.1]),
    expected_probs=torch.tensor([0.0693098, 0.0990131, 0.100677, 0.075837, 0.0838128, 0.0926275, 0.102369, 0.113135,
                                 0.125034, 0.138184, 0.0703223, 0.0921197, 0.0958792, 0.0769448, 0.0850372, 0.0939806,
                                 0.103865, 0.114788, 0.126861, 0.140203, 0.080888, 0.12942, 0.110285, 0.0885056,
                                 0.0978138, 0.108101, 0.11947, 0.0885056, 0.0885056, 0.0885056, 0.0715989, 0.0895156,
                                 0.0837425, 0.0783417, 0.0865809, 0.0956867, 0.10575, 0.116872, 0.129164, 0.142748]).reshape(4, 10),
    atol=1e-3)


# CudaSamplerTest.cc:842-903 (testBanRepeatNGram): token histories [4, 9], index of the last token per row, n-gram sizes, the one
# token each row must lose
NGRAM = dict(
    logits=torch.tensor([0.1, 0.1, 0.1, 0.1, 0.2, 0.3, 0.1, 0.1, 0.1, 0.1] * 4).reshape(4, 10),
    token_ids=torch.tensor([0, 2, 3, 4, 5, 0, 0, 2, 0, 1, 2, 3, 3, 3, 1, 2, 3, 0, 1, 2, 1, 2, 1, 2, 1, 2, 0, 9, 8, 6, 9, 8, 0, 0, 0, 0],
                           dtype=torch.int32).reshape(4, 9),
    sequence_last_index=torch.tensor([7, 7, 7, 4], dtype=torch.int32), no_repeat_ngram_size=torch.tensor([3, 4, 5, 2], dtype=torch.int32),
    banned=[3, 3, 1, 6])


def oracle_sample_greedy(oracle, c, uniform):
    """The sampler flow (CudaSampleOp.cc:619-800) composed from the oracle's pieces.  -> (ids, probabilities after filter)."""
    step = c["step"]
    x = c["logits"].clone()
    history = c["token_ids"].t().contiguous()
    if bool((c["temperature"] != 1).any()):
        x = oracle.apply_penalties(x, temperature=c["temperature"])
    if "repetition_penalty" in c:
        lengths = c["input_lengths"].clone()
        lengths[: c["sequence_lengths"].numel()] = c["sequence_lengths"]
        x = oracle.apply_penalties(x, repetition_penalty=c["repetition_penalty"], presence_penalty=c["presence_penalty"],
                                   frequency_penalty=c["frequency_penalty"], output_ids=history, input_lengths=lengths,
                                   max_input_length=step + 1, step=step + 1)
    probs = oracle.softmax_rows(x)
    if bool((c["top_k"] == 1).all()):
        return probs.argmax(-1).int(), probs
    filtered = oracle.top_k_top_p_filter(probs, c["top_k"], c["top_p"])
    return oracle.sample_rows(filtered, uniform), filtered
