"""Known-answer vectors for chain rejection sampling, transcribed from the reference's own kernel tests
(rtp_llm/models_py/bindings/cuda/test/CudaSpeculativeSamplingTest.cc:36-366): inputs -> (output_token_ids, accepted)."""
import torch


def _z(*shape):
    return torch.zeros(*shape, dtype=torch.float32)


def cases():
    out = []
    # RejectionSampling_AllAccept (:36-88)
    B, G, V = 2, 3, 16
    dp, tp = _z(B, G, V), _z(B, G + 1, V)
    dp[:, :, 5] = 1.0; tp[:, :, 5] = 1.0
    out.append(dict(name="all_accept", draft_probs=dp, target_probs=tp, draft_ids=torch.full((B, G), 5, dtype=torch.int32),
                    target_ids=torch.full((B, G + 1), 5, dtype=torch.int32), uniform=_z(B, G + 1), do_sample=torch.ones(B, dtype=torch.bool),
                    expect_ids=[[5, 5, 5, 5], [5, 5, 5, 5]], expect_acc=[4, 4]))
    # RejectionSampling_ImmediateReject (:90-141), greedy row: direct fallback to the target token
    B, G = 1, 3
    dp, tp = _z(B, G, V), _z(B, G + 1, V)
    dp[:, :, 3] = 1.0; tp[:, :, 7] = 1.0
    out.append(dict(name="immediate_reject", draft_probs=dp, target_probs=tp, draft_ids=torch.full((B, G), 3, dtype=torch.int32),
                    target_ids=torch.full((B, G + 1), 7, dtype=torch.int32), uniform=torch.full((B, G + 1), 0.5),
                    do_sample=torch.zeros(B, dtype=torch.bool), expect_ids=[[7, -1, -1, -1]], expect_acc=[1]))
    # RejectionSampling_GreedyTargetUsesDirectFallbackWithFullQ (:143-184)
    B, G = 1, 1
    dp, tp = _z(B, G, V), _z(B, G + 1, V)
    dp[0, 0, 3] = 0.4; dp[0, 0, 7] = 0.6; tp[0, 0, 7] = 0.51; tp[0, 0, 8] = 0.49
    out.append(dict(name="greedy_direct_fallback", draft_probs=dp, target_probs=tp, draft_ids=torch.full((B, G), 3, dtype=torch.int32),
                    target_ids=torch.full((B, G + 1), 7, dtype=torch.int32), uniform=torch.full((B, G + 1), 0.99),
                    do_sample=torch.zeros(B, dtype=torch.bool), expect_ids=[[7, -1]], expect_acc=[1]))
    # RejectionSampling_PartialAccept (:186-244)
    B, G = 1, 3
    dp, tp = _z(B, G, V), _z(B, G + 1, V)
    dp[:, :, 5] = 1.0; dp[:, 2, 5] = 0.0; dp[:, 2, 3] = 1.0; tp[:, :, 7] = 1.0
    di = torch.full((B, G), 5, dtype=torch.int32); di[:, 2] = 3
    ti = torch.full((B, G + 1), 5, dtype=torch.int32); ti[:, 2] = 7; ti[:, 3] = 7
    out.append(dict(name="partial_accept", draft_probs=dp, target_probs=tp, draft_ids=di, target_ids=ti, uniform=torch.full((B, G + 1), 0.5),
                    do_sample=torch.zeros(B, dtype=torch.bool), expect_ids=[[5, 5, 7, -1]], expect_acc=[3]))
    # RejectionSampling_StochasticSameTokenStillUsesRatio (:246-288)
    B, G = 1, 1
    dp, tp = _z(B, G, V), _z(B, G + 1, V)
    dp[0, 0, 5] = 0.9; dp[0, 0, 7] = 0.1; tp[0, 0, 5] = 0.2; tp[0, 0, 7] = 0.8
    out.append(dict(name="stochastic_same_token", draft_probs=dp, target_probs=tp, draft_ids=torch.full((B, G), 5, dtype=torch.int32),
                    target_ids=torch.full((B, G + 1), 5, dtype=torch.int32), uniform=torch.full((B, G + 1), 0.5),
                    do_sample=torch.ones(B, dtype=torch.bool), expect_ids=[[7, -1]], expect_acc=[1]))
    # RejectionSampling_PointMassStochasticSameTokenCanReject (:290-327): draft_probs = nullptr, point mass
    tp = _z(B, G + 1, V); tp[0, 0, 5] = 0.2; tp[0, 0, 7] = 0.8
    out.append(dict(name="point_mass_same_token", draft_probs=None, target_probs=tp, draft_ids=torch.full((B, G), 5, dtype=torch.int32),
                    target_ids=torch.full((B, G + 1), 5, dtype=torch.int32), uniform=torch.full((B, G + 1), 0.5),
                    do_sample=torch.ones(B, dtype=torch.bool), expect_ids=[[7, -1]], expect_acc=[1]))
    # RejectionSampling_ImplicitPointMassDraft (:329-366)
    out.append(dict(name="implicit_point_mass", draft_probs=None, target_probs=tp.clone(), draft_ids=torch.full((B, G), 5, dtype=torch.int32),
                    target_ids=torch.full((B, G + 1), 7, dtype=torch.int32), uniform=torch.full((B, G + 1), 0.5),
                    do_sample=torch.ones(B, dtype=torch.bool), expect_ids=[[7, -1]], expect_acc=[1]))
    return out


def random_case(B, G, V, seed, point_mass=False):
    """Random rows on a dyadic grid (probabilities are multiples of 2^-12, uniforms of 2^-10): every fp32 partial sum is
    exact in any order, so kernel and oracle must agree bit for bit."""
    g = torch.Generator().manual_seed(seed)

    def dist(*lead):
        # sparse support of 32 tokens: 31 weights in [1, 64] and one that tops the row up to exactly 4096
        n = 1
        for d in lead:
            n *= d
        w = torch.zeros(n, V, dtype=torch.int64)
        for r in range(n):
            idx = torch.randperm(V, generator=g)[: min(32, V)]
            vals = torch.randint(1, 65, (len(idx),), generator=g)
            vals[-1] = 4096 - vals[:-1].sum()
            w[r, idx] = vals
        assert (w >= 0).all() and (w.sum(-1) == 4096).all()
        return (w.float() / 4096.0).reshape(*lead, V)

    tp = dist(B, G + 1)
    dp = None if point_mass else dist(B, G)
    src = tp[:, :G] if point_mass else dp
    draft_ids = torch.multinomial(src.reshape(-1, V) + 1e-9, 1, generator=g).reshape(B, G).int()
    target_ids = tp.argmax(-1).int()
    flip = torch.rand(B, G, generator=g) < 0.5               # make about half the greedy positions agree
    target_ids[:, :G] = torch.where(flip, draft_ids, target_ids[:, :G])
    uniform = torch.randint(0, 1024, (B, G + 1), generator=g).float() / 1024.0
    do_sample = torch.rand(B, generator=g) < 0.6
    return dict(draft_probs=dp, target_probs=tp, draft_ids=draft_ids, target_ids=target_ids, uniform=uniform, do_sample=do_sample)
