"""GPU parity for the speculative-verify row: the rejection-sampling / softmax kernels against the
oracle (bit-exact token ids), and end-to-end losslessness of greedy speculative decoding on the decode path."""
import pytest
import torch

import spec_vectors
from oracle import oracle
from rtp_llm_amd import _C, model, ops
from rtp_llm_amd.speculative import SpeculativeDecoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_native():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _C.lib()


def _run(case):
    dp = None if case["draft_probs"] is None else case["draft_probs"].to(DEV)
    out, acc = ops.rejection_sample(case["draft_ids"].to(DEV), case["target_ids"].to(DEV), case["target_probs"].to(DEV),
                                    case["uniform"].to(DEV), case["do_sample"].to(DEV), dp)
    torch.cuda.synchronize()
    return out.cpu(), acc.cpu()


@pytest.mark.parametrize("case", spec_vectors.cases(), ids=lambda c: c["name"])
def test_rejection_sample_reference_known_answers(case):
    out, acc = _run(case)
    assert out.tolist() == case["expect_ids"] and acc.tolist() == case["expect_acc"]


@pytest.mark.parametrize("B,G,V", [(5, 1, 97), (8, 4, 1000), (3, 7, 4099), (2, 3, 152064)])
@pytest.mark.parametrize("point_mass", [False, True])
def test_rejection_sample_matches_oracle_bit_exact(B, G, V, point_mass):
    case = spec_vectors.random_case(B, G, V, 100 + B + G, point_mass)
    out, acc = _run(case)
    ref_out, ref_acc = oracle.rejection_sample(case["draft_ids"], case["target_ids"], case["target_probs"], case["uniform"],
                                               case["do_sample"], case["draft_probs"])
    assert torch.equal(acc, ref_acc) and torch.equal(out, ref_out)


def test_rejection_sample_token_stride_and_errors():
    case = spec_vectors.random_case(4, 3, 257, 9)
    wide = torch.full((4, 4, 3), -7, dtype=torch.int32)          # [B, G+1, stride]: only the last element counts
    wide[..., 2] = case["target_ids"]
    a = _run(case)
    b = _run({**case, "target_ids": wide})
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with pytest.raises(_C.Mi355Error):
        ops.rejection_sample(case["draft_ids"].to(DEV), case["target_ids"].to(DEV), case["target_probs"][:, :2].contiguous().to(DEV),
                             case["uniform"].to(DEV), case["do_sample"].to(DEV))


@pytest.mark.parametrize("R,V,T", [(3, 1000, 1.0), (10, 152064, 0.7)])
def test_softmax_rows(R, V, T):
    x = torch.randn(R, V, generator=torch.Generator().manual_seed(R)) * 4
    p = ops.softmax_rows(x.to(DEV), T).cpu()
    ref = oracle.softmax_rows(x, T)
    assert torch.allclose(p, ref, atol=1e-6, rtol=1e-4) and torch.allclose(p.sum(-1), torch.ones(R), atol=1e-4)


def _engines(seed_t, seed_d, same):
    cfg_t = model.ModelConfig("tiny-target", 3, 512, 8, 2, 64, 1024, 2048, max_pos=512)
    cfg_d = cfg_t if same else model.ModelConfig("tiny-draft", 1, 256, 4, 2, 64, 512, 2048, max_pos=512)
    wt = model.synth_model(cfg_t, "fp16" if same else "w4", DEV, seed=seed_t)
    wd = wt if same else model.synth_model(cfg_d, "fp16", DEV, seed=seed_d)
    mk = lambda cfg, w, mb: model.DecoderEngine(cfg, w, kv_int8=False, page=16, num_blocks=64, max_batch=mb, max_seq_len=128, device=DEV)
    return cfg_t, mk(cfg_t, wt, 32), mk(cfg_d, wd, 16), mk(cfg_t, wt, 8)


@pytest.mark.parametrize("same_draft", [False, True])
def test_greedy_speculative_decoding_is_lossless(same_draft):
    """Tokens emitted by propose + verify equal plain greedy decoding of the target (the verify rows are decode rows with
    their own position: causal masking by construction).  A random draft is rejected almost always (1 token / round); a
    draft identical to the target is accepted always (gamma + 1 tokens / round)."""
    B, G, prompt_len, gen = 3, 4, 6, 20
    cfg, target, draft, plain = _engines(3, 11, same_draft)
    g = torch.Generator().manual_seed(5)
    prompt = torch.randint(0, cfg.vocab, (B, prompt_len), generator=g, dtype=torch.int32)
    bt = torch.arange(B * 8, dtype=torch.int32).reshape(B, 8)
    # "prefill" through the decode path: feed the prompt token by token to all three engines
    for eng in (target, draft, plain):
        for s in range(prompt_len - 1):
            eng.set_inputs(prompt[:, s].tolist(), [s] * B, bt)
            eng.forward(B)
    # plain greedy reference stream
    plain.set_inputs(prompt[:, -1].tolist(), [prompt_len - 1] * B, bt)
    ref = []
    for _ in range(gen):
        plain.step(B)
        ref.append(plain.token_ids[:B].cpu().clone())
    ref = torch.stack(ref, 1).tolist()
    spec = SpeculativeDecoder(target, draft, G)
    spec.start(prompt[:, -1].tolist(), [prompt_len - 1] * B, bt, bt)
    got = [[] for _ in range(B)]
    rounds = 0
    while min(len(x) for x in got) < gen:
        for b, toks in enumerate(spec.step()):
            got[b] += toks
        rounds += 1
    for b in range(B):
        assert got[b][:gen] == ref[b], (b, got[b][:gen], ref[b])
    if same_draft:
        assert all(a == G + 1 for r in spec.accepted_hist for a in r) and rounds == -(-gen // (G + 1))
    else:
        assert rounds > gen // (G + 1)


@pytest.mark.parametrize("R,V", [(7, 97), (4, 4099), (3, 152064)])
def test_sample_rows_matches_oracle_bit_exact(R, V):
    """Dyadic probabilities (multiples of 2^-12) and uniforms (2^-10): fp32 prefix sums are exact in any order."""
    case = spec_vectors.random_case(R, 1, V, 7 + R)
    probs, u = case["target_probs"][:, 0].contiguous(), case["uniform"][:, 0].contiguous()
    got = ops.sample_rows(probs.to(DEV), u.to(DEV)).cpu()
    assert torch.equal(got, oracle.sample_rows(probs, u))


def test_sampled_speculative_rows_preserve_the_target_distribution():
    """Rejection sampling fed with the target's SAMPLED tokens (as the reference's SpeculativeSampler does): the first
    emitted token of a sampled row is distributed as the target's first-position distribution q0 whatever the draft
    proposes -- accepted draft with probability min(1, q0[d]/p0[d]), residual draw otherwise, and q-samples as bonus.
    Statistical check over 60k independent rows on a 12-token vocabulary (chi-square-like bound, 5 sigma)."""
    R, G, V = 60000, 2, 12
    g = torch.Generator().manual_seed(17)
    q = torch.softmax(torch.randn(G + 1, V, generator=g) * 1.5, -1)
    pd = torch.softmax(torch.randn(G, V, generator=g) * 1.5, -1)
    tp = q.unsqueeze(0).expand(R, G + 1, V).contiguous()
    dp = pd.unsqueeze(0).expand(R, G, V).contiguous()
    draft_ids = torch.stack([torch.multinomial(pd[i], R, replacement=True, generator=g) for i in range(G)], 1).int()
    u = torch.rand(R, G + 1, generator=g)
    ut = torch.rand(R * (G + 1), generator=g)
    target_ids = ops.sample_rows(tp.reshape(-1, V).to(DEV), ut.to(DEV)).reshape(R, G + 1)
    out, acc = ops.rejection_sample(draft_ids.to(DEV), target_ids, tp.to(DEV), u.to(DEV),
                                    torch.ones(R, dtype=torch.bool, device=DEV), dp.to(DEV))
    first = out[:, 0].cpu().long()
    emp = torch.bincount(first, minlength=V).float() / R
    sigma = torch.sqrt(q[0] * (1 - q[0]) / R)
    assert bool(((emp - q[0]).abs() < 5 * sigma + 1e-4).all()), (emp, q[0])
    # bonus token (all drafts accepted): distributed as q[G]
    full = acc.cpu() == G + 1
    n = int(full.sum())
    assert n > 2000
    empb = torch.bincount(out[:, G].cpu().long()[full], minlength=V).float() / n
    sb = torch.sqrt(q[G] * (1 - q[G]) / n)
    assert bool(((empb - q[G]).abs() < 5 * sb + 1e-4).all()), (empb, q[G])


def test_speculative_refuses_to_outgrow_the_cache():
    cfg, target, draft, _ = _engines(3, 11, False)
    spec = SpeculativeDecoder(target, draft, 4)
    bt = torch.arange(3 * 8, dtype=torch.int32).reshape(3, 8)       # 8 blocks x 16 = 128 tokens
    with pytest.raises(_C.Mi355Error):
        spec.start([1, 2, 3], [10, 125, 3], bt, bt)                  # 125 + gamma + 1 > 128
