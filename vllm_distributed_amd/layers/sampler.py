"""Token sampling.

Order of operations on fp32 logits (reference: vllm/v1/sample/sampler.py:18-227):
raw logprobs (if requested) -> allowed-token mask -> bad words -> min-tokens
stop masking -> logit bias -> penalties -> one fused HIP kernel that applies
temperature / top-k / top-p / min-p and draws (greedy rows take the argmax).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..request import LogprobsLists


@dataclass
class SamplingMetadata:
    temperature: Optional[torch.Tensor]         # [S] float32; None = all greedy
    all_greedy: bool
    all_random: bool
    top_k: Optional[torch.Tensor] = None        # [S] int32
    top_p: Optional[torch.Tensor] = None        # [S] float32
    min_p: Optional[torch.Tensor] = None
    seeds: Optional[torch.Tensor] = None        # [S] int64 (per-step derived)
    max_num_logprobs: Optional[int] = None
    no_penalties: bool = True
    prompt_token_ids: Optional[torch.Tensor] = None    # [S, Lp] padded with vocab_size
    output_token_ids: list[list[int]] = field(default_factory=list)
    frequency_penalties: Optional[torch.Tensor] = None
    presence_penalties: Optional[torch.Tensor] = None
    repetition_penalties: Optional[torch.Tensor] = None
    min_tokens: dict[int, tuple[int, set[int]]] = field(default_factory=dict)   # row -> (min, stop ids)
    logit_bias: list[Optional[dict[int, float]]] = field(default_factory=list)
    allowed_token_ids_mask: Optional[torch.Tensor] = None     # [S, V] bool, True = forbidden
    bad_words_token_ids: dict[int, list[list[int]]] = field(default_factory=dict)


@dataclass
class SamplerOutput:
    sampled_token_ids: torch.Tensor           # [S, 1] (or [S, k+1] from the rejection sampler, -1 padded)
    logprobs_tensors: Optional[tuple] = None  # (token_ids [S, k+1], logprobs [S, k+1], ranks [S])


def token_counts(token_lists: list[list[int]], vocab: int, device) -> torch.Tensor:
    """[S, V] int32 occurrence counts."""
    S = len(token_lists)
    counts = torch.zeros((S, vocab + 1), dtype=torch.int32, device=device)
    L = max((len(t) for t in token_lists), default=0)
    if L == 0:
        return counts[:, :vocab]
    pad = torch.full((S, L), vocab, dtype=torch.long)
    for i, t in enumerate(token_lists):
        if t:
            pad[i, : len(t)] = torch.as_tensor(t, dtype=torch.long)
    pad = pad.to(device, non_blocking=True)
    counts.scatter_add_(1, pad, torch.ones_like(pad, dtype=torch.int32))
    return counts[:, :vocab]


class Sampler(nn.Module):
    def forward(self, logits: torch.Tensor, md: SamplingMetadata) -> SamplerOutput:
        logits = logits.float()
        raw = None
        if md.max_num_logprobs is not None:
            raw = torch.log_softmax(logits, dim=-1)
        logits = self.apply_logits_processors(logits, md)
        sampled = self.sample(logits, md)
        lp = None
        if md.max_num_logprobs is not None:
            lp = self.gather_logprobs(raw, md.max_num_logprobs, sampled)
        return SamplerOutput(sampled.unsqueeze(-1), lp)

    # ------------------------------------------------------------------
    def apply_logits_processors(self, logits: torch.Tensor, md: SamplingMetadata) -> torch.Tensor:
        if md.allowed_token_ids_mask is not None:
            logits.masked_fill_(md.allowed_token_ids_mask, float("-inf"))
        if md.bad_words_token_ids:
            self._apply_bad_words(logits, md)
        if md.min_tokens:
            for row, (mn, stops) in md.min_tokens.items():
                if len(md.output_token_ids[row]) < mn and stops:
                    logits[row, list(stops)] = float("-inf")
        if md.logit_bias and any(md.logit_bias):
            rows, cols, vals = [], [], []
            for i, lb in enumerate(md.logit_bias):
                if lb:
                    for t, b in lb.items():
                        rows.append(i); cols.append(t); vals.append(b)
            if rows:
                logits[rows, cols] += torch.tensor(vals, dtype=logits.dtype, device=logits.device)
        if not md.no_penalties:
            V = logits.shape[-1]
            out_counts = token_counts(md.output_token_ids, V, logits.device)
            pm = torch.zeros((logits.shape[0], V + 1), dtype=torch.bool, device=logits.device)
            if md.prompt_token_ids is not None:
                pm.scatter_(1, md.prompt_token_ids.long().clamp(max=V), True)
            ops.apply_penalties(logits, pm[:, :V].contiguous(), out_counts.contiguous(),
                                md.repetition_penalties, md.frequency_penalties, md.presence_penalties)
        return logits

    @staticmethod
    def _apply_bad_words(logits, md):
        for row, words in md.bad_words_token_ids.items():
            past = md.output_token_ids[row]
            for w in words:
                if len(w) - 1 > len(past):
                    continue
                prefix = w[:-1]
                if not prefix or past[-len(prefix):] == prefix:
                    logits[row, w[-1]] = float("-inf")

    def sample(self, logits: torch.Tensor, md: SamplingMetadata) -> torch.Tensor:
        if md.all_greedy or md.temperature is None:
            return ops.argmax(logits)
        return ops.sample(logits, md.temperature, md.top_k, md.top_p, md.min_p, md.seeds)

    @staticmethod
    def gather_logprobs(logprobs: torch.Tensor, k: int, token_ids: torch.Tensor):
        """(ids [S, k+1], logprobs [S, k+1], rank of the sampled token [S]); column 0 is the sampled token."""
        tok = token_ids.long().unsqueeze(-1)
        tok_lp = logprobs.gather(-1, tok)
        ranks = (logprobs >= tok_lp).sum(-1)
        if k > 0:
            top_lp, top_ids = torch.topk(logprobs, k, dim=-1)
            ids = torch.cat((tok, top_ids), dim=1)
            lps = torch.cat((tok_lp, top_lp), dim=1)
        else:
            ids, lps = tok, tok_lp
        return ids.to(torch.int32), lps, ranks

    @staticmethod
    def to_lists(lp: Optional[tuple]) -> Optional[LogprobsLists]:
        if lp is None:
            return None
        ids, lps, ranks = lp
        return LogprobsLists(ids.tolist(), lps.tolist(), ranks.tolist())
