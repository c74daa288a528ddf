"""Verification of speculative draft tokens.

``logits`` holds, for every request, the target model's logits at its draft positions plus one
bonus position (len(draft)+1 rows, requests concatenated).  Greedy requests accept a draft token
iff it equals the target arg-max.  Sampled requests use the standard speculative-sampling rule:
accept draft token d with probability min(1, p(d)/q(d)), on rejection draw from the normalised
residual max(p - q, 0); draft sources without probabilities (n-gram, greedy heads) count as
q = one-hot(d).  Everything is computed for all rows at once on the device; the accepted prefix is
cut on the host from a single read-back.
Reference: vllm/v1/sample/rejection_sampler.py (:19-365 and its Triton kernels :368-631).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from .sampler import Sampler, SamplingMetadata


def _expand(t: Optional[torch.Tensor], counts: torch.Tensor) -> Optional[torch.Tensor]:
    return None if t is None else torch.repeat_interleave(t, counts, dim=0)


class RejectionSampler:
    def __call__(self, logits: torch.Tensor, drafts: list[list[int]], md: SamplingMetadata, sampler: Sampler,
                 draft_probs: Optional[torch.Tensor] = None) -> list[list[int]]:
        dev = logits.device
        S = len(drafts)
        counts_l = [len(d) + 1 for d in drafts]
        R = sum(counts_l)
        assert logits.shape[0] == R, (logits.shape, R)
        counts = torch.tensor(counts_l, dtype=torch.int64, device=dev)
        logits = logits.float()
        # per-request processors see the same output history for every row of a request
        if md.allowed_token_ids_mask is not None or md.logit_bias or not md.no_penalties or md.min_tokens \
                or md.bad_words_token_ids:
            logits = sampler.apply_logits_processors(logits, self._expand_md(md, counts_l, counts))
        # draft token per row (bonus rows: -1)
        flat_draft = []
        for d in drafts:
            flat_draft.extend(d)
            flat_draft.append(-1)
        draft_t = torch.tensor(flat_draft, dtype=torch.int64, device=dev)
        has_draft = draft_t >= 0
        safe_draft = draft_t.clamp(min=0)
        greedy_tok = logits.argmax(dim=-1)
        if md.all_greedy or md.temperature is None:
            accept = (greedy_tok == draft_t) & has_draft
            fallback = greedy_tok
        else:
            temp = _expand(md.temperature, counts)
            is_greedy = temp < 1e-5
            scaled = logits / temp.clamp(min=1e-5).unsqueeze(1)
            if md.top_k is not None or md.top_p is not None or md.min_p is not None:
                scaled = ops.topk_topp_mask(scaled, _expand(md.top_k, counts), _expand(md.top_p, counts),
                                            _expand(md.min_p, counts))
            probs = torch.softmax(scaled, dim=-1)
            gen = None
            if md.seeds is not None:
                gen = torch.Generator(device=dev)
                gen.manual_seed(int(md.seeds.sum().item()) & 0x7FFFFFFFFFFFFFFF)
            u = torch.rand(R, device=dev, generator=gen, dtype=torch.float32)
            p_d = probs.gather(1, safe_draft.unsqueeze(1)).squeeze(1)
            if draft_probs is not None:
                q_d = draft_probs.gather(1, safe_draft.unsqueeze(1)).squeeze(1).clamp(min=1e-10)
                residual = (probs - draft_probs).clamp(min=0)
            else:
                q_d = torch.ones_like(p_d)
                residual = probs.clone()
                residual[has_draft, safe_draft[has_draft]] = 0
            accept_rand = (u * q_d < p_d) & has_draft
            # bonus rows draw from the full distribution, rejected rows from the residual
            dist = torch.where(has_draft.unsqueeze(1), residual, probs)
            empty = dist.sum(dim=-1) <= 0           # residual can vanish when p == q
            dist[empty] = probs[empty]
            noise = torch.empty_like(dist).exponential_(1.0, generator=gen)
            drawn = (dist / noise).argmax(dim=-1)
            accept = torch.where(is_greedy, (greedy_tok == draft_t) & has_draft, accept_rand)
            fallback = torch.where(is_greedy, greedy_tok, drawn)
        acc, fb = accept.tolist(), fallback.tolist()
        out: list[list[int]] = []
        r = 0
        for i, d in enumerate(drafts):
            toks = []
            for j, tok in enumerate(d):
                if acc[r + j]:
                    toks.append(tok)
                else:
                    toks.append(fb[r + j])
                    break
            else:
                toks.append(fb[r + len(d)])      # every draft accepted: bonus token
            out.append(toks)
            r += len(d) + 1
        return out

    @staticmethod
    def _expand_md(md: SamplingMetadata, counts_l: list[int], counts: torch.Tensor) -> SamplingMetadata:
        rows_of = []
        for i, c in enumerate(counts_l):
            rows_of.extend([i] * c)
        e = SamplingMetadata(temperature=None, all_greedy=md.all_greedy, all_random=md.all_random)
        e.no_penalties = md.no_penalties
        e.prompt_token_ids = _expand(md.prompt_token_ids, counts)
        e.frequency_penalties = _expand(md.frequency_penalties, counts)
        e.presence_penalties = _expand(md.presence_penalties, counts)
        e.repetition_penalties = _expand(md.repetition_penalties, counts)
        e.allowed_token_ids_mask = _expand(md.allowed_token_ids_mask, counts)
        if md.output_token_ids:
            e.output_token_ids = [md.output_token_ids[i] for i in rows_of]
        if md.logit_bias:
            e.logit_bias = [md.logit_bias[i] for i in rows_of]
        first = {}
        r = 0
        for i, c in enumerate(counts_l):
            first[i] = r
            r += c
        e.min_tokens = {first[i] + j: v for i, v in md.min_tokens.items() for j in range(counts_l[i])}
        e.bad_words_token_ids = {first[i] + j: v for i, v in md.bad_words_token_ids.items()
                                 for j in range(counts_l[i])}
        return e
