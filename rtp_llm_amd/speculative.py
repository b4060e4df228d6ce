"""Speculative decoding over the decode path: separate draft model proposes, the target verifies.

The reference's "vanilla" mode (cpp/config/ConfigModules.h:305-312; wiring
cpp/engine_base/ProposeModelEngineInitParams.h:16-30): a small draft model (BASELINE config 5: Qwen2-0.5B) proposes
`gamma` tokens, the target scores gamma+1 tokens per sequence in ONE step over the paged cache (`is_target_verify`,
models_py/bindings/OpDefs.h:283) and chain rejection sampling keeps the longest accepted prefix plus one corrected /
bonus token (cpp/normal_engine/speculative/SpeculativeSampler.cc:214, kernel
bindings/rocm/speculative_sampling/sampling.cu:306).

MI355X design: a verify row (sequence b, draft position t) is a query row with its own position ctx_b + t:
`mi355_rope_kv_write_rows` stores the K/V of all gamma+1 tokens first, `mi355_paged_attn_rows` then lets row t see
ctx_b + t + 1 tokens -- the causal mask lives inside the page walk, and the gamma+1 rows of a sequence ride in the MFMA
column dimension next to its GQA heads, so the sequence's KV is streamed once, not gamma+1 times.
The linears see M = B * (gamma + 1) rows (<= 64: the wide-batch GEMM shapes), so verifying 5 tokens costs about one
decode step of a 5x larger batch.  Rejected tokens leave stale K/V beyond the accepted length; they are overwritten
by the next step's tokens at the same positions.
"""
from typing import List, Optional, Sequence

import torch

from . import _C, ops
from .model import DecoderEngine


class SpeculativeDecoder:
    """Greedy or sampled speculative decoding on two DecoderEngines that share nothing but the token stream.  For rows
    with do_sample set the target's own token at every verify position is SAMPLED from its distribution (the reference
    hands rejection sampling the target sampler's outputs, SpeculativeSampler.cc:214), so the bonus token after gamma
    accepted drafts is a sample too and the output distribution is the target's; greedy rows use the argmax.

    State per sequence: `last` (newest token, not yet in the target cache), `ctx` (tokens in the target cache), and for
    the draft cache `dctx` plus the list of tokens it has not processed yet (`pending`, ends with `last`)."""

    def __init__(self, target: DecoderEngine, draft: DecoderEngine, gamma: int):
        if gamma < 1:
            raise ValueError("gamma >= 1")
        self.target, self.draft, self.gamma = target, draft, gamma
        self.B = 0
        self.profile = None      # set to {} to collect per-phase wall times (synchronises at every phase boundary)

    def start(self, last_tokens: Sequence[int], ctx_lens: Sequence[int], target_block_table: torch.Tensor,
              draft_block_table: torch.Tensor, draft_ctx_lens: Optional[Sequence[int]] = None,
              draft_pending: Optional[List[List[int]]] = None):
        """Both caches already hold `ctx_lens[b]` tokens of sequence b (or `draft_ctx_lens` + `draft_pending` when the
        draft lags); `last_tokens[b]` is the next input token."""
        B = len(last_tokens)
        if B * (self.gamma + 1) > self.target.max_batch or 2 * B > self.draft.max_batch:
            raise _C.Mi355Error(f"speculative: batch {B} x (gamma+1) exceeds the engines' max_batch")
        self.B = B
        self.last = [int(t) for t in last_tokens]
        self.ctx = [int(c) for c in ctx_lens]
        self.dctx = [int(c) for c in (draft_ctx_lens if draft_ctx_lens is not None else ctx_lens)]
        self.pending = [list(p) for p in draft_pending] if draft_pending is not None else [[t] for t in self.last]
        self.tbt = torch.as_tensor(target_block_table, dtype=torch.int32)
        self.dbt = torch.as_tensor(draft_block_table, dtype=torch.int32)
        self.accepted_hist: List[List[int]] = []
        self._check_room()

    def _check_room(self):
        """A round writes positions ctx .. ctx + gamma of every sequence in both caches."""
        self.target.check_room(self.ctx, self.gamma + 1, self.tbt, "speculative verify")
        self.draft.check_room([d + len(p) for d, p in zip(self.dctx, self.pending)], self.gamma, self.dbt, "speculative draft")

    # one engine forward over `rows` = [(sequence, token, position)], logits stay in engine.logits[:len(rows)]
    @staticmethod
    def _forward(eng: DecoderEngine, rows, table: torch.Tensor):
        seqs = [r[0] for r in rows]
        eng.set_inputs([r[1] for r in rows], [r[2] for r in rows], table[seqs])
        eng.forward(len(rows))

    def step(self, do_sample: Optional[Sequence[bool]] = None, temperature: float = 1.0,
             uniform: Optional[torch.Tensor] = None, uniform_target: Optional[torch.Tensor] = None) -> List[List[int]]:
        """One propose + verify round; returns the tokens emitted per sequence (1 .. gamma + 1 each).
        uniform [B, gamma+1]: accept / residual draws; uniform_target [B, gamma+1]: the target's own samples (sampled rows)."""
        B, G = self.B, self.gamma
        dev = self.target.device
        self._check_room()
        import time
        t_last = [time.perf_counter()]

        def mark(name):
            if self.profile is not None:
                torch.cuda.synchronize()
                now = time.perf_counter()
                self.profile[name] = self.profile.get(name, 0.0) + (now - t_last[0]) * 1e3
                t_last[0] = now
        # ---- draft: catch up on the pending tokens (1 or 2 per sequence), then gamma - 1 single-token steps
        rows, last_row = [], []
        for b in range(B):
            for k, tok in enumerate(self.pending[b]):
                rows.append((b, tok, self.dctx[b] + k))
            last_row.append(len(rows) - 1)
            self.dctx[b] += len(self.pending[b])
        self._forward(self.draft, rows, self.dbt)
        nxt = ops.argmax(self.draft.logits[: len(rows)])[torch.tensor(last_row, device=dev)]
        drafts = [nxt]
        mark("draft_catch_up")
        if G > 1:
            # the remaining gamma - 1 draft steps never leave the device: one token per sequence, greedy feedback and
            # position increment inside the captured step (DecoderEngine.replay), no host round trip per token
            self.draft.token_ids[:B].copy_(nxt)
            self.draft.positions[:B].copy_(torch.tensor(self.dctx, dtype=torch.int32))
            self.draft.block_table[:B, : self.dbt.shape[1]].copy_(self.dbt)
            self.draft.capture(B)
            for _ in range(G - 1):
                self.draft.replay(B, 1)
                drafts.append(self.draft.token_ids[:B].clone())
            for b in range(B):
                self.dctx[b] += G - 1
        draft_ids = torch.stack(drafts, dim=1).to(torch.int32).contiguous()          # [B, G]
        dl = draft_ids.tolist()
        mark("draft_steps")
        # ---- target: gamma + 1 decode rows per sequence = causal verify over the paged cache
        toks, poss = [], []
        for b in range(B):
            toks += [self.last[b]] + dl[b]
            poss += [self.ctx[b] + t for t in range(G + 1)]
        # gamma + 1 consecutive rows per sequence, one block-table row per sequence: the multi-row attention kernel walks each
        # sequence's KV once for all of its rows (is_target_verify)
        self.target.set_inputs(toks, poss, self.tbt[:B])
        self.target.forward(B * (G + 1), q_len=G + 1)
        R = B * (G + 1)
        mark("target_verify")
        logits = self.target.logits[:R]
        target_ids = ops.argmax(logits).reshape(B, G + 1).contiguous()
        probs = ops.softmax_rows(logits, temperature).reshape(B, G + 1, -1)
        ds = torch.zeros(B, dtype=torch.uint8, device=dev) if do_sample is None else torch.as_tensor(do_sample, device=dev).to(torch.uint8)
        if uniform is None:
            uniform = torch.rand(B, G + 1, device=dev, dtype=torch.float32)
        if bool(ds.any()):   # sampled rows: the target's token at each position is a draw from its distribution
            if uniform_target is None:
                uniform_target = torch.rand(B, G + 1, device=dev, dtype=torch.float32)
            sampled = ops.sample_rows(probs.reshape(R, -1), uniform_target.to(dev).reshape(R).contiguous()).reshape(B, G + 1)
            target_ids = torch.where(ds.bool().unsqueeze(1), sampled, target_ids).contiguous()
        out, acc = ops.rejection_sample(draft_ids, target_ids, probs, uniform.to(dev), ds)   # draft = point mass (greedy draft)
        out_l, acc_l = out.tolist(), acc.tolist()
        mark("sample")
        # ---- advance: target cache now holds `last` and the accepted drafts; the draft cache holds last + d_1..d_{G-1}
        emitted = []
        for b in range(B):
            n = acc_l[b]
            toks = out_l[b][:n]
            emitted.append(toks)
            old_ctx = self.ctx[b]
            self.ctx[b] += n
            self.last[b] = toks[-1]
            valid_draft = old_ctx + 1 + min(n - 1, G - 1)          # last_old + accepted drafts the draft model has seen
            self.dctx[b] = valid_draft
            self.pending[b] = ([dl[b][G - 1]] if n - 1 == G else []) + [toks[-1]]
        self.accepted_hist.append(acc_l)
        return emitted
