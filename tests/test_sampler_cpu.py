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
