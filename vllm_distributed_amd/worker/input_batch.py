"""Persistent per-request state on the worker.

The scheduler only sends deltas (new requests in full, known requests as new
block ids + counters), so the worker keeps token ids, block tables and sampling
parameters of every live request in flat numpy arrays indexed by a batch row;
building one step's inputs is then a handful of vectorised gathers.

Reference: vllm/v1/worker/gpu_input_batch.py, block_table.py (same role; the
layout and the per-step path are different: rows are never reordered here, the
step order is an index array, which also gives the decode-first ordering).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from ..request import LoRARequest
from ..sampling_params import SamplingParams, SamplingType


@dataclass
class CachedRequestState:
    req_id: str
    row: int
    sampling_params: Optional[SamplingParams]
    pooling_params: object = None
    lora_request: Optional[LoRARequest] = None
    generator_seed: Optional[int] = None
    num_prompt_tokens: int = 0
    output_token_ids: list[int] = field(default_factory=list)
    mm_inputs: list = field(default_factory=list)
    mm_positions: list = field(default_factory=list)


class InputBatch:
    def __init__(self, max_num_reqs: int, max_model_len: int, block_size: int, vocab_size: int,
                 num_kv_groups: int = 1, pin_memory: bool = False):
        self.max_num_reqs = max_num_reqs
        self.max_model_len = max_model_len
        self.block_size = block_size
        self.vocab_size = vocab_size
        self.max_blocks = -(-max_model_len // block_size)
        self.num_kv_groups = num_kv_groups
        self.token_ids = np.zeros((max_num_reqs, max_model_len), dtype=np.int32)
        self.num_tokens = np.zeros(max_num_reqs, dtype=np.int32)          # tokens known (prompt + sampled)
        self.num_computed = np.zeros(max_num_reqs, dtype=np.int32)
        self.block_tables = [np.zeros((max_num_reqs, self.max_blocks), dtype=np.int32)
                             for _ in range(num_kv_groups)]
        self.num_blocks = np.zeros((num_kv_groups, max_num_reqs), dtype=np.int32)
        # sampling parameters
        self.temperature = np.zeros(max_num_reqs, dtype=np.float32)
        self.top_k = np.zeros(max_num_reqs, dtype=np.int32)
        self.top_p = np.ones(max_num_reqs, dtype=np.float32)
        self.min_p = np.zeros(max_num_reqs, dtype=np.float32)
        self.freq_pen = np.zeros(max_num_reqs, dtype=np.float32)
        self.pres_pen = np.zeros(max_num_reqs, dtype=np.float32)
        self.rep_pen = np.ones(max_num_reqs, dtype=np.float32)
        self.seed = np.zeros(max_num_reqs, dtype=np.int64)
        self.has_seed = np.zeros(max_num_reqs, dtype=bool)
        self.num_logprobs = np.full(max_num_reqs, -1, dtype=np.int32)
        self.num_prompt_logprobs = np.full(max_num_reqs, -1, dtype=np.int32)
        self.lora_id = np.zeros(max_num_reqs, dtype=np.int32)
        self.reqs: dict[str, CachedRequestState] = {}
        self.row_to_req: list[Optional[str]] = [None] * max_num_reqs
        self.free_rows = list(range(max_num_reqs - 1, -1, -1))

    @property
    def num_reqs(self) -> int:
        return len(self.reqs)

    # ---- lifecycle -----------------------------------------------------------
    def add_request(self, req_id: str, prompt_token_ids: list[int], sampling_params, block_ids, num_computed: int,
                    pooling_params=None, lora_request=None, mm_inputs=None, mm_positions=None,
                    output_token_ids: Optional[list[int]] = None) -> CachedRequestState:
        if not self.free_rows:
            raise RuntimeError("input batch is full")
        row = self.free_rows.pop()
        st = CachedRequestState(req_id, row, sampling_params, pooling_params, lora_request,
                                num_prompt_tokens=len(prompt_token_ids), mm_inputs=mm_inputs or [],
                                mm_positions=mm_positions or [])
        self.reqs[req_id] = st
        self.row_to_req[row] = req_id
        n = len(prompt_token_ids)
        self.token_ids[row, :n] = prompt_token_ids
        if output_token_ids:
            self.token_ids[row, n:n + len(output_token_ids)] = output_token_ids
            st.output_token_ids = list(output_token_ids)
            n += len(output_token_ids)
        self.num_tokens[row] = n
        self.num_computed[row] = num_computed
        self.num_blocks[:, row] = 0
        self.append_blocks(row, block_ids)
        sp = sampling_params
        if sp is not None:
            greedy = sp.sampling_type == SamplingType.GREEDY
            self.temperature[row] = 0.0 if greedy else sp.temperature
            self.top_k[row] = sp.top_k if 0 < sp.top_k < self.vocab_size else 0
            self.top_p[row] = sp.top_p
            self.min_p[row] = sp.min_p
            self.freq_pen[row], self.pres_pen[row], self.rep_pen[row] = \
                sp.frequency_penalty, sp.presence_penalty, sp.repetition_penalty
            self.has_seed[row] = sp.seed is not None
            self.seed[row] = sp.seed if sp.seed is not None else 0
            self.num_logprobs[row] = sp.logprobs if sp.logprobs is not None else -1
            self.num_prompt_logprobs[row] = sp.prompt_logprobs if sp.prompt_logprobs is not None else -1
        else:
            self.temperature[row] = 0.0
            self.num_logprobs[row] = self.num_prompt_logprobs[row] = -1
        self.lora_id[row] = lora_request.lora_int_id if lora_request else 0
        return st

    def remove_request(self, req_id: str) -> None:
        st = self.reqs.pop(req_id, None)
        if st is None:
            return
        self.row_to_req[st.row] = None
        self.free_rows.append(st.row)

    def append_blocks(self, row: int, block_ids) -> None:
        for g, ids in enumerate(block_ids):
            if not ids:
                continue
            s = int(self.num_blocks[g, row])
            self.block_tables[g][row, s:s + len(ids)] = ids
            self.num_blocks[g, row] = s + len(ids)

    def reset_blocks(self, row: int, block_ids) -> None:
        self.num_blocks[:, row] = 0
        self.append_blocks(row, block_ids)

    def append_token(self, row: int, token: int) -> None:
        n = int(self.num_tokens[row])
        self.token_ids[row, n] = token
        self.num_tokens[row] = n + 1
