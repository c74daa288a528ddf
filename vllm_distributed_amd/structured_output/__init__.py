"""Structured (guided) decoding manager; backends in structured_output/backends.py.
Reference: vllm/v1/structured_output/__init__.py."""
from __future__ import annotations

from concurrent.futures import Future, ThreadPoolExecutor
from typing import Optional

import numpy as np

from ..logger import init_logger

logger = init_logger(__name__)


class StructuredOutputRequest:
    def __init__(self, params, future: Optional[Future] = None):
        self.params = params
        self._future = future
        self._grammar = None

    @property
    def ready(self) -> bool:
        if self._grammar is not None:
            return True
        if self._future is not None and self._future.done():
            self._grammar = self._future.result()
            return True
        return False

    @property
    def grammar(self):
        if self._grammar is None and self._future is not None:
            self._grammar = self._future.result()
        return self._grammar

    def accept_tokens(self, request_id: str, tokens: list[int]) -> bool:
        return self.grammar.accept_tokens(tokens)

    def validate_tokens(self, tokens: list[int]) -> list[int]:
        return self.grammar.validate_tokens(tokens)


class StructuredOutputManager:
    def __init__(self, vllm_config):
        self.vllm_config = vllm_config
        self._pool: Optional[ThreadPoolExecutor] = None
        self._backend = None
        self._tokenizer = None

    def _get_backend(self):
        if self._backend is None:
            from ..engine.processor import get_tokenizer
            from .backends import FSMBackend
            mc = self.vllm_config.model_config
            self._tokenizer = get_tokenizer(mc.tokenizer, mc.tokenizer_mode)
            self._backend = FSMBackend(self._tokenizer, mc.get_vocab_size(),
                                       self.vllm_config.decoding_config.disable_any_whitespace)
            self._pool = ThreadPoolExecutor(max_workers=2, thread_name_prefix="grammar")
        return self._backend

    def make_request(self, engine_core_request) -> StructuredOutputRequest:
        gd = engine_core_request.sampling_params.guided_decoding
        backend = self._get_backend()
        fut = self._pool.submit(backend.compile, gd)
        return StructuredOutputRequest(gd, fut)

    def grammar_bitmask(self, requests: dict, structured_ids: dict[str, int], spec_tokens: dict) -> Optional[np.ndarray]:
        """int32 [num_structured_rows, ceil(V/32)], bit set = token allowed; row order follows
        ``structured_ids`` values sorted ascending (re-indexed in the returned dict order)."""
        if not structured_ids:
            return None
        V = self.vllm_config.model_config.get_vocab_size()
        words = (V + 31) // 32
        order = sorted(structured_ids, key=structured_ids.get)
        out = np.full((len(order), words), -1, dtype=np.int32)
        for j, rid in enumerate(order):
            g = requests[rid].structured_output_request.grammar
            g.fill_bitmask(out[j])
            structured_ids[rid] = j
        return out

    def shutdown(self) -> None:
        if self._pool is not None:
            self._pool.shutdown(wait=False, cancel_futures=True)
