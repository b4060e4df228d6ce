"""Oracle of the sampler's non-greedy branch against the reference's own known answers (CPU)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import sampler_vectors as sv


@pytest.mark.parametrize("name", sorted(sv.CASES))
def test_oracle_sampler_reproduces_reference_known_answers(name):
    c = sv.case(name)
    seen = [set() for _ in range(4)]
    for u in np.linspace(0.0, 0.999, 41, dtype=np.float32):
        ids, probs = sv.oracle_sample_greedy(oracle, c, torch.full((4,), float(u)))
        for r in range(4):
            assert int(ids[r]) in c["allowed"][r], (name, r, float(u), int(ids[r]))
            seen[r].add(int(ids[r]))
        assert torch.allclose(probs.sum(-1), torch.ones(4), atol=1e-5)
    if name != "top_k_1":   # the accepted sets are the filter's support: a sweep over u reaches every member
        for r in range(4):
            assert seen[r] == c["allowed"][r], (name, r, seen[r])


def test_oracle_penalties_reproduce_reference_probabilities():
    c = sv.PENALTY
    _, probs = sv.oracle_sample_greedy(oracle, c, torch.zeros(4))
    assert torch.allclose(probs, c["expected_probs"], atol=c["atol"], rtol=0), (probs - c["expected_probs"]).abs().max()


def test_oracle_penalties_skip_padding_and_foreign_ids():
    x = torch.zeros(2, 8) + 1.0
    ids = torch.tensor([[3, 9], [3, -1], [5, 2], [6, 2]], dtype=torch.int32)      # [step = 4, batch = 2]; 9 and -1 are outside the vocabulary
    out = oracle.apply_penalties(x, repetition_penalty=torch.tensor([2.0, 1.0]), presence_penalty=torch.tensor([0.0, 0.5]),
                                 frequency_penalty=torch.tensor([0.0, 0.25]), output_ids=ids, input_lengths=torch.tensor([2, 4]),
                                 max_input_length=3, step=4)
    # row 0: positions [2, 3) are padding -> ids {3, 3, 6}; row 1: ids {2, 2}
    assert out[0].tolist() == [1, 1, 1, 0.5, 1, 1, 0.5, 1]
    assert out[1].tolist() == [1, 1, 0.0, 1, 1, 1, 1, 1]


def test_oracle_filter_definitions():
    p = torch.tensor([[0.05, 0.4, 0.25, 0.2, 0.1]])
    f = oracle.top_k_top_p_filter(p, torch.tensor([3]), torch.tensor([1.0]))
    assert torch.allclose(f, torch.tensor([[0, 0.4, 0.25, 0.2, 0]]) / 0.85)
    f = oracle.top_k_top_p_filter(p, torch.tensor([0]), torch.tensor([0.5]))      # mass before 0.25 is 0.4 <= 0.5: kept; before 0.2 is 0.65: dropped
    assert torch.allclose(f, torch.tensor([[0, 0.4, 0.25, 0, 0]]) / 0.65)
    f = oracle.top_k_top_p_filter(p, torch.tensor([0]), torch.tensor([0.0]))      # 0 reads as 1
    assert torch.allclose(f, p)


class _OracleOps:
    """Stand-in for rtp_llm_amd.ops inside THIS test only: the host flow of sampler.sample_greedy is plain Python and is checked
    here without a GPU by letting the oracle play the kernels (and by recording which kernels the flow asked for)."""

    def __init__(self):
        self.calls = []

    def apply_penalties(self, logits, **kw):
        self.calls.append("apply_penalties:" + ",".join(sorted(k for k, v in kw.items() if v is not None and k.endswith(("temperature", "penalty")))))
        logits.copy_(oracle.apply_penalties(logits, **kw))
        return logits

    def argmax(self, x):
        self.calls.append("argmax")
        return x.argmax(-1).int()

    def softmax_rows(self, x, temperature=1.0):
        self.calls.append("softmax_rows")
        return oracle.softmax_rows(x, temperature)

    def top_k_top_p_sample(self, probs, top_k, top_p, uniform, return_probs=False):
        self.calls.append("top_k_top_p_sample")
        f = oracle.top_k_top_p_filter(probs, top_k, top_p)
        ids = oracle.sample_rows(f, uniform)
        return (ids, f) if return_probs else ids


@pytest.mark.parametrize("name", sorted(sv.CASES))
def test_sample_greedy_host_flow(name, monkeypatch):
    """Order of operations and host-side shortcuts of sampleGreedy (CudaSampleOp.cc:633-700,739-786)."""
    from rtp_llm_amd import sampler
    fake = _OracleOps()
    monkeypatch.setattr(sampler, "ops", fake)
    c = sv.case(name)
    u = torch.full((4,), 0.37)
    p = sampler.GreedyParams(logits=c["logits"].clone(), input_lengths=c["input_lengths"], sequence_lengths=c["sequence_lengths"],
                             token_ids=c["token_ids"].clone(), step=c["step"], top_k=c["top_k"], top_p=c["top_p"].clone(),
                             temperature=c["temperature"], uniform=u)
    ids = sampler.sample_greedy(p)
    want, _ = sv.oracle_sample_greedy(oracle, c, u)
    assert torch.equal(ids, want.int()) and torch.equal(p.token_ids[:, c["step"]], ids)
    if name == "top_k_1":      # temperature, then arg-max of the logits: no softmax
        assert fake.calls == ["apply_penalties:temperature", "argmax"]
    else:
        assert fake.calls == ["apply_penalties:temperature", "softmax_rows", "top_k_top_p_sample"]
    assert torch.equal(c["top_p"], sv.case(name)["top_p"])          # the caller's top_p is not edited in place


def test_sample_greedy_host_flow_penalties_and_probabilities(monkeypatch):
    from rtp_llm_amd import sampler
    fake = _OracleOps()
    monkeypatch.setattr(sampler, "ops", fake)
    c = sv.PENALTY
    probs = torch.zeros(4, 10)
    p = sampler.GreedyParams(logits=c["logits"].clone(), input_lengths=c["input_lengths"], sequence_lengths=c["sequence_lengths"],
                             token_ids=c["token_ids"].clone(), step=c["step"], top_k=c["top_k"], top_p=c["top_p"], temperature=c["temperature"],
                             repetition_penalty=c["repetition_penalty"], presence_penalty=c["presence_penalty"],
                             frequency_penalty=c["frequency_penalty"], output_all_probs=probs, uniform=torch.full((4,), 0.5))
    sampler.sample_greedy(p)
    # every T == 1: no temperature launch; penalties in one call; the probabilities the reference's test expects
    assert fake.calls == ["apply_penalties:frequency_penalty,presence_penalty,repetition_penalty", "softmax_rows", "top_k_top_p_sample"]
    assert torch.allclose(probs, c["expected_probs"], atol=c["atol"], rtol=0)
    # default penalties: the launch is skipped altogether
    fake.calls.clear()
    p2 = sampler.GreedyParams(logits=c["logits"].clone(), input_lengths=c["input_lengths"], sequence_lengths=c["sequence_lengths"],
                              token_ids=c["token_ids"].clone(), step=c["step"], top_k=torch.ones(4, dtype=torch.int32), top_p=c["top_p"],
                              temperature=c["temperature"], repetition_penalty=torch.ones(4), presence_penalty=torch.zeros(4),
                              frequency_penalty=torch.zeros(4))
    sampler.sample_greedy(p2)
    assert fake.calls == ["argmax"]
    with pytest.raises(ValueError):
        sampler.sample_greedy(sampler.GreedyParams(logits=c["logits"].clone(), input_lengths=c["input_lengths"],
                                                   sequence_lengths=c["sequence_lengths"], token_ids=c["token_ids"].clone(), step=c["step"],
                                                   top_k=c["top_k"], top_p=c["top_p"], temperature=c["temperature"],
                                                   repetition_penalty=torch.ones(4)))


def test_oracle_ban_repeat_ngram_reproduces_reference_known_answer():
    c = sv.NGRAM
    out = oracle.ban_repeat_ngram(c["logits"], c["token_ids"], c["sequence_last_index"], c["no_repeat_ngram_size"])
    for r in range(4):
        for j in range(10):
            assert (out[r, j] == float("-inf")) == (j == c["banned"][r]), (r, j)
            assert j == c["banned"][r] or out[r, j] == c["logits"][r, j]
    # n = 0 and sequences shorter than n: untouched; n = 1 bans every token seen so far
    out = oracle.ban_repeat_ngram(c["logits"], c["token_ids"], torch.tensor([7, 1, 7, 4]), torch.tensor([0, 4, 1, 1]))
    assert torch.equal(out[:2], c["logits"][:2])
    assert sorted(torch.nonzero(out[2] == float("-inf")).flatten().tolist()) == [1, 2]
    assert sorted(torch.nonzero(out[3] == float("-inf")).flatten().tolist()) == [6, 8, 9]


def _threshold_model(probs, top_k, top_p):
    """The kernel's algorithm (csrc/sampling.hip top_k_top_p_sample_kernel) restated with exact arithmetic: thresholds on the
    fp32 bit pattern -- kb = k-th largest pattern, vb = the smallest pattern whose strictly-larger mass fits top_p -- and, for the
    entries equal to vb, as many in index order as S + j v <= p allows.  Checked against the reference-shaped oracle (sort +
    cumsum) so that the sort-free formulation itself is pinned on the CPU."""
    out = []
    for r in range(probs.shape[0]):
        q = probs[r].numpy().astype(np.float32)
        V = len(q)
        b = q.view(np.uint32).astype(np.int64)
        k = int(top_k[r])
        tp = np.float32(top_p[r])
        if abs(tp) < 1e-7:
            tp = np.float32(1)
        keep = np.ones(V, dtype=bool)
        if 0 < k < V:
            keep &= b >= np.sort(b)[::-1][k - 1]
        if abs(tp - 1) >= 1e-7 and q.sum(dtype=np.float64) > tp:
            vb = next(t for t in np.unique(b) if q[b > t].sum(dtype=np.float64) <= tp)
            S = q[b > vb].sum(dtype=np.float64)
            v = float(np.array([vb], dtype=np.uint32).view(np.float32)[0])
            kp = b > vb
            rank = 0
            for j in range(V):
                if b[j] == vb:
                    if S + rank * v > tp:
                        break
                    kp[j] = True
                    rank += 1
            keep &= kp
        f = np.where(keep, q, 0).astype(np.float32)
        out.append(f / np.float32(max(f.sum(dtype=np.float32), 1e-10)))
    return torch.from_numpy(np.stack(out))


@pytest.mark.parametrize("V,levels", [(50, 4095), (640, 5), (3000, 3), (4095, 4095)])
def test_sort_free_threshold_formulation_equals_sort_and_cumsum(V, levels):
    g = torch.Generator().manual_seed(V + levels)
    R = 6
    if levels >= V:        # distinct values
        probs = torch.stack([(torch.randperm(levels, generator=g)[:V] + 1).float() * 2.0 ** -22 for _ in range(R)])
    else:                  # few distinct values: long runs of equals, the boundary class is kept partially
        probs = torch.randint(1, levels + 1, (R, V), generator=g).float() * 2.0 ** -20
    total = probs.sum(-1)
    top_p = (torch.tensor([0.05, 0.3, 0.5, 0.77, 0.999, 1.0]) * total).float()
    top_p[-1] = 1.0
    top_k = torch.tensor([0, 0, V // 3, 7, V + 5, 1], dtype=torch.int32)
    want = oracle.top_k_top_p_filter(probs, top_k, top_p)
    assert torch.equal(_threshold_model(probs, top_k, top_p), want)
