"""The sampler's non-greedy branch on the GPU (mi355_apply_penalties, mi355_top_k_top_p_sample, sampler.sample_greedy) against the
reference's known answers and the oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle
from rtp_llm_amd import _C, ops, sampler
from tests import sampler_vectors as sv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(c, uniform, **extra):
    return sampler.GreedyParams(logits=c["logits"].clone().to(DEV), input_lengths=c["input_lengths"], sequence_lengths=c["sequence_lengths"],
                                token_ids=c["token_ids"].clone(), step=c["step"], top_k=c["top_k"], top_p=c["top_p"].clone(),
                                temperature=c["temperature"], uniform=uniform, **extra)


@pytest.mark.parametrize("name", sorted(sv.CASES))
def test_sampler_reproduces_reference_known_answers(name):
    """CudaSamplerTest.cc:518-780: exact tokens for the greedy rows, membership in the accepted set for the sampled ones -- and,
    for every uniform, the token the oracle draws."""
    c = sv.case(name)
    seen = [set() for _ in range(4)]
    for u in np.linspace(0.0, 0.999, 21, dtype=np.float32):
        uni = torch.full((4,), float(u))
        p = _params(c, uni)
        ids = sampler.sample_greedy(p).cpu()
        want, _ = sv.oracle_sample_greedy(oracle, c, uni)
        assert torch.equal(ids, want.int()), (name, float(u), ids, want)
        assert torch.equal(p.token_ids[:, c["step"]], ids) and torch.equal(p.token_ids[:, : c["step"]], c["token_ids"][:, : c["step"]])
        for r in range(4):
            assert int(ids[r]) in c["allowed"][r]
            seen[r].add(int(ids[r]))
    if name != "top_k_1":
        assert seen == c["allowed"]


def test_penalties_reproduce_reference_probabilities():
    """CudaSamplerTest.cc:905-975: the distribution after repetition / presence / frequency penalties, to the reference's 1e-3."""
    c = sv.PENALTY
    probs = torch.zeros(4, 10, device=DEV)
    p = _params(c, torch.full((4,), 0.5), repetition_penalty=c["repetition_penalty"], presence_penalty=c["presence_penalty"],
                frequency_penalty=c["frequency_penalty"], output_all_probs=probs)
    ids = sampler.sample_greedy(p).cpu()
    assert torch.allclose(probs.cpu(), c["expected_probs"], atol=c["atol"], rtol=0)
    want, wprobs = sv.oracle_sample_greedy(oracle, c, torch.full((4,), 0.5))
    assert torch.equal(ids, want.int()) and torch.allclose(probs.cpu(), wprobs, atol=1e-6)


@pytest.mark.parametrize("B,V,step", [(3, 97, 7), (5, 4099, 300), (4, 152064, 2500), (2, 1031, 40000)])   # the last: beyond 32 history entries per lane
def test_apply_penalties_matches_oracle(B, V, step):
    g = torch.Generator().manual_seed(B * 1000 + step)
    x = torch.randn(B, V, generator=g) * 3
    ids = torch.randint(-2, V + 2, (step, B), generator=g, dtype=torch.int32)       # a few ids outside the vocabulary
    ids[: step // 2, 0] = 5                                                          # a heavily repeated id
    lens = torch.randint(1, step, (B,), generator=g, dtype=torch.int32)
    max_in = step - 2
    temp = torch.rand(B, generator=g) + 0.5
    rep, pres, freq = torch.rand(B, generator=g) + 0.8, torch.rand(B, generator=g), torch.rand(B, generator=g) * 0.1
    got = ops.apply_penalties(x.clone().to(DEV), temperature=temp, repetition_penalty=rep, presence_penalty=pres, frequency_penalty=freq,
                              output_ids=ids.to(DEV), input_lengths=lens, max_input_length=max_in, step=step).cpu()
    ref = oracle.apply_penalties(oracle.apply_penalties(x, temperature=temp), repetition_penalty=rep, presence_penalty=pres,
                                 frequency_penalty=freq, output_ids=ids, input_lengths=lens, max_input_length=max_in, step=step)
    assert torch.equal(got, ref)                # same fp32 operations in the same order (no contraction on either side)
    # temperature alone; penalties alone with the optional vectors absent
    assert torch.equal(ops.apply_penalties(x.clone().to(DEV), temperature=temp).cpu(), oracle.apply_penalties(x, temperature=temp))
    got = ops.apply_penalties(x.clone().to(DEV), presence_penalty=pres, output_ids=ids.to(DEV), max_input_length=0, step=step).cpu()
    assert torch.equal(got, oracle.apply_penalties(x, presence_penalty=pres, output_ids=ids, max_input_length=0, step=step))


def _distinct_dyadic(R, V, seed):
    """Rows of DISTINCT multiples of 2^-22 below 2^-10: every partial sum is exact in fp32 in any order, no ties."""
    g = torch.Generator().manual_seed(seed)
    assert V <= 4095
    return torch.stack([(torch.randperm(4095, generator=g)[:V] + 1).float() * 2.0 ** -22 for _ in range(R)])


@pytest.mark.parametrize("R,V", [(6, 50), (5, 1000), (3, 4095)])
def test_top_k_top_p_bit_exact_on_distinct_dyadic_rows(R, V):
    probs = _distinct_dyadic(R, V, R * 7 + V)
    g = torch.Generator().manual_seed(V)
    total = probs.sum(-1)
    top_k = torch.randint(0, V + 3, (R,), generator=g, dtype=torch.int32)
    top_k[0] = 0
    top_p = (torch.rand(R, generator=g) * total).float()       # the rows do not sum to 1: p is a mass, whatever the total
    top_p[-1] = 1.0
    u = (torch.randint(0, 1024, (R,), generator=g).float() / 1024)
    ids, out = ops.top_k_top_p_sample(probs.to(DEV), top_k, top_p, u.to(DEV), return_probs=True)
    want = oracle.top_k_top_p_filter(probs, top_k, top_p)
    assert torch.equal(out.cpu(), want)
    assert torch.equal(ids.cpu(), oracle.sample_rows(torch.where(want > 0, probs, torch.zeros(())), u))   # the draw runs on the unnormalised survivors
    # no filter at all == sample_rows
    ids0 = ops.top_k_top_p_sample(probs.to(DEV), None, None, u.to(DEV))
    assert torch.equal(ids0.cpu(), oracle.sample_rows(probs, u))


def test_top_k_top_p_full_vocabulary_rows():
    """Softmax rows at the Qwen2 vocabulary: top_p placed in the middle of a gap of the exact (float64) cumulative masses, so that the
    kept set does not depend on fp32 summation order; the kept sets must agree and the probabilities to 1e-6 relative."""
    R, V = 4, 152064
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(torch.randn(R, V, generator=g) * 4, -1)
    sp = probs.double().sort(-1, descending=True).values
    cs = sp.cumsum(-1)
    cut = [40, 300, 5, 2000]                                        # entries kept by top-p
    top_p = torch.tensor([float((cs[r, cut[r] - 2] + cs[r, cut[r] - 1]) / 2) for r in range(R)])
    top_k = torch.tensor([0, 100, 50, 0], dtype=torch.int32)
    u = torch.tensor([0.1, 0.5, 0.9, 0.99])
    ids, out = ops.top_k_top_p_sample(probs.to(DEV), top_k, top_p, u.to(DEV), return_probs=True)
    want = oracle.top_k_top_p_filter(probs, top_k, top_p)
    out = out.cpu()
    assert torch.equal(out > 0, want > 0)
    assert [(int((out[r] > 0).sum())) for r in range(R)] == [40, 100, 5, 2000]
    assert torch.allclose(out, want, rtol=1e-5, atol=0)
    for r in range(R):          # the drawn index is a survivor whose cumulative-mass interval (float64, index order) contains u
        j = int(ids[r])
        kept = torch.where(want[r] > 0, probs[r], torch.zeros(())).double()
        c = kept.cumsum(0) / kept.sum()
        assert kept[j] > 0 and float(c[j] - kept[j] / kept.sum()) - 1e-5 <= float(u[r]) <= float(c[j]) + 1e-5


def test_ties_with_the_kth_value_stay_and_top_p_takes_equal_values_in_index_order():
    probs = torch.tensor([[0.1, 0.3, 0.3, 0.2, 0.1], [0.25, 0.25, 0.25, 0.125, 0.125], [0.125, 0.25, 0.125, 0.25, 0.25]])
    top_k, top_p = torch.tensor([1, 0, 0], dtype=torch.int32), torch.tensor([1.0, 0.3, 0.5])
    ids, out = ops.top_k_top_p_sample(probs.to(DEV), top_k, top_p, torch.zeros(3, device=DEV), return_probs=True)
    # row 1: the mass before the occurrences of 0.25 is 0, 0.25, 0.5 -> two stay; row 2: 0, 0.25, 0.5 <= 0.5 -> all three stay
    assert torch.allclose(out.cpu(), torch.tensor([[0, 0.5, 0.5, 0, 0], [0.5, 0.5, 0, 0, 0], [0, 1 / 3, 0, 1 / 3, 1 / 3]]))
    assert torch.equal(out.cpu(), oracle.top_k_top_p_filter(probs, top_k, top_p))
    assert ids.cpu().tolist() == [1, 0, 1]


@pytest.mark.parametrize("V", [640, 152064])
def test_top_p_boundary_inside_a_long_run_of_equal_values(V):
    """Dyadic rows with few distinct values (every class has many members, spread over many thread segments): exact sums, so the
    number of boundary-class members that stay must match the stable-sort oracle exactly."""
    g = torch.Generator().manual_seed(V)
    R = 4
    probs = (torch.randint(1, 6, (R, V), generator=g).float() * 2.0 ** -20)
    total = probs.sum(-1)
    top_p = torch.tensor([0.1, 0.37, 0.62, 0.93]) * total
    top_k = torch.tensor([0, 0, V // 3, 0], dtype=torch.int32)
    u = torch.tensor([0.0, 0.25, 0.5, 0.999])
    ids, out = ops.top_k_top_p_sample(probs.to(DEV), top_k, top_p, u.to(DEV), return_probs=True)
    want = oracle.top_k_top_p_filter(probs, top_k, top_p)
    assert torch.equal(out.cpu() > 0, want > 0)
    assert torch.equal(out.cpu(), want)
    assert torch.equal(ids.cpu(), oracle.sample_rows(torch.where(want > 0, probs, torch.zeros(())), u))


def test_sampler_refuses_missing_uniform_and_cpu_logits():
    c = sv.case("top_p")
    with pytest.raises(ValueError):
        sampler.sample_greedy(_params(c, None))
    p = _params(c, torch.zeros(4))
    p.logits = p.logits.cpu()
    with pytest.raises(_C.Mi355Error):
        sampler.sample_greedy(p)


def test_ban_repeat_ngram_reproduces_reference_known_answer_and_oracle():
    """CudaSamplerTest.cc:842-903, then random histories (short vocabularies make repeats frequent) against the oracle, bit for bit."""
    c = sv.NGRAM
    got = ops.ban_repeat_ngram(c["logits"].clone().to(DEV), c["token_ids"].to(DEV), c["sequence_last_index"], c["no_repeat_ngram_size"]).cpu()
    for r in range(4):
        assert torch.nonzero(got[r] == float("-inf")).flatten().tolist() == [c["banned"][r]]
    assert torch.equal(got, oracle.ban_repeat_ngram(c["logits"], c["token_ids"], c["sequence_last_index"], c["no_repeat_ngram_size"]))
    g = torch.Generator().manual_seed(12)
    B, V, L = 9, 37, 700
    x = torch.randn(B + 2, V, generator=g)                                  # two extra rows (context rows): untouched
    tok = torch.randint(-1, 6, (B, L), generator=g, dtype=torch.int32)      # ids in [-1, 5]: many repeats, a few invalid ids
    last = torch.randint(0, L, (B,), generator=g, dtype=torch.int32)
    last[0], last[1] = L - 1, 0
    ng = torch.tensor([2, 3, 0, 1, 4, 2, 7, 3, 700], dtype=torch.int32)
    got = ops.ban_repeat_ngram(x.clone().to(DEV), tok.to(DEV), last, ng).cpu()
    # the oracle indexes the logits with the banned id: feed it only the rows' valid ids (an invalid id is skipped by the kernel)
    exp = x.clone()
    for b in range(B):
        n, N = int(ng[b]), int(last[b]) + 1
        if n == 0 or N < n:
            continue
        t = tok[b, :N].tolist()
        for i in range(N - n + 1):
            if t[i:i + n - 1] == t[N - n + 1:N] and 0 <= t[i + n - 1] < V:
                exp[b, t[i + n - 1]] = float("-inf")
    assert torch.equal(got, exp)
    # through the sampler flow: greedy rows with a 2-gram ban never repeat a bigram
    p = sampler.GreedyParams(logits=c["logits"].clone().to(DEV), input_lengths=torch.full((4,), -1, dtype=torch.int32),
                             sequence_lengths=c["sequence_last_index"] + 1, token_ids=c["token_ids"].clone(), step=8,
                             top_k=torch.ones(4, dtype=torch.int32), top_p=torch.ones(4), temperature=torch.ones(4),
                             no_repeat_ngram_size=c["no_repeat_ngram_size"])
    ids = sampler.sample_greedy(p).cpu()
    masked = oracle.ban_repeat_ngram(c["logits"], c["token_ids"], c["sequence_last_index"], c["no_repeat_ngram_size"])
    assert torch.equal(ids, masked.argmax(-1).int())


def test_generate_sampled_end_to_end():
    """Engine + sampler: top_k = 1 and a one-token nucleus reproduce greedy generation token for token; a seeded sampled run is
    reproducible; a 2-gram ban leaves no repeated bigram; a presence penalty of 1e4 never emits a token twice."""
    from rtp_llm_amd import model
    cfg = model.ModelConfig("tiny-sampler", 2, 512, 8, 2, 64, 1024, 2048, max_pos=512)
    w = model.synth_model(cfg, "w4", DEV, seed=3)
    mk = lambda: model.DecoderEngine(cfg, w, kv_int8=False, page=16, num_blocks=64, max_batch=8, max_seq_len=128, device=DEV)
    prompts = [[5, 9, 2, 7], [11, 3], [8, 8, 8, 1, 4, 6]]
    bt = torch.arange(3 * 8, dtype=torch.int32).reshape(3, 8)
    greedy = mk().generate(prompts, bt, 12)
    assert mk().generate_sampled(prompts, bt, 12, top_k=1) == greedy
    assert mk().generate_sampled(prompts, bt, 12, top_k=0, top_p=1e-6, temperature=0.7) == greedy
    a = mk().generate_sampled(prompts, bt, 12, top_k=8, top_p=0.9, temperature=1.3, seed=5)
    assert a == mk().generate_sampled(prompts, bt, 12, top_k=8, top_p=0.9, temperature=1.3, seed=5)
    assert a != mk().generate_sampled(prompts, bt, 12, top_k=8, top_p=0.9, temperature=1.3, seed=6)
    ng = mk().generate_sampled(prompts, bt, 24, top_k=1, no_repeat_ngram_size=2)
    for b in range(3):
        seq = prompts[b] + ng[b]
        for j in range(len(prompts[b]), len(seq)):            # no generated token completes a bigram that occurred before
            assert (seq[j - 1], seq[j]) not in set(zip(seq[: j - 1], seq[1:j])), (b, j, seq)
    pp = mk().generate_sampled(prompts, bt, 16, top_k=1, presence_penalty=1e4)
    for b in range(3):
        assert len(set(pp[b])) == 16 and not set(pp[b]) & set(prompts[b]), (b, pp[b])
