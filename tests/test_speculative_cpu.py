"""CPU suite for the speculative-verify row: the oracle's rejection sampling against the reference's own
known-answer tests, invariants on random cases, and the no-op contract of the C-ABI entry point."""
import ctypes as C

import pytest
import torch

from oracle import oracle
from rtp_llm_amd import _C
import spec_vectors


@pytest.mark.parametrize("case", spec_vectors.cases(), ids=lambda c: c["name"])
def test_oracle_matches_reference_known_answers(case):
    ids, acc = oracle.rejection_sample(case["draft_ids"], case["target_ids"], case["target_probs"], case["uniform"],
                                       case["do_sample"], case["draft_probs"])
    assert ids.tolist() == case["expect_ids"]
    assert acc.tolist() == case["expect_acc"]


@pytest.mark.parametrize("point_mass", [False, True])
def test_oracle_invariants_on_random_rows(point_mass):
    c = spec_vectors.random_case(6, 4, 97, 5, point_mass)
    ids, acc = oracle.rejection_sample(c["draft_ids"], c["target_ids"], c["target_probs"], c["uniform"], c["do_sample"], c["draft_probs"])
    G = c["draft_ids"].shape[1]
    for b in range(ids.shape[0]):
        n = int(acc[b])
        assert 1 <= n <= G + 1
        assert ids[b, : n - 1].tolist() == c["draft_ids"][b, : n - 1].tolist()      # accepted prefix = the drafts
        assert (ids[b, n:] == -1).all() and ids[b, n - 1] >= 0                       # one correction / bonus token, then padding
        if not c["do_sample"][b]:                                                    # greedy rows emit the target's tokens
            assert ids[b, n - 1] == c["target_ids"][b, n - 1]
            assert all(c["draft_ids"][b, i] == c["target_ids"][b, i] for i in range(n - 1))
        elif n <= G:                                                                 # a resampled token has residual mass
            q = c["target_probs"][b, n - 1]
            p = torch.zeros_like(q)
            if point_mass:
                p[c["draft_ids"][b, n - 1]] = 1.0
            else:
                p = c["draft_probs"][b, n - 1]
            assert (q - p)[ids[b, n - 1]] > 0


def test_batch_size_zero_is_a_noop_like_the_reference():
    # invokeRejectionSampling returns success for batch 0 with null pointers (CudaSpeculativeSamplingTest.cc:368-372)
    assert _C.lib().mi355_rejection_sample(None, None, None, None, None, 1, None, None, None, 0, 3, 16, 0, None) == 0


def test_rejects_missing_pointers():
    assert _C.lib().mi355_rejection_sample(None, None, None, None, None, 1, None, None, None, 2, 3, 16, 0, None) < 0
    assert b"rejection_sample" in _C.lib().mi355_last_error()
