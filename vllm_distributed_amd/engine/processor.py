"""Input side of the front end: tokenizer loading, parameter validation and the
construction of ``EngineCoreRequest`` (reference: vllm/v1/engine/processor.py,
vllm/transformers_utils/tokenizer.py, vllm/inputs/preprocess.py)."""
from __future__ import annotations

import json
import os
import time
from typing import Any, Optional, Union

from ..config import VllmConfig
from ..inputs import PromptType, parse_prompt
from ..logger import init_logger
from ..request import EngineCoreRequest, LoRARequest
from ..sampling_params import PoolingParams, SamplingParams

logger = init_logger(__name__)


def get_tokenizer(path: str, tokenizer_mode: str = "auto", trust_remote_code: bool = False, **kw):
    """Tokenizer from a local directory (there is no hub access)."""
    if not os.path.isdir(path):
        raise FileNotFoundError(f"tokenizer directory {path!r} not found; pass skip_tokenizer_init=True and "
                                "token-id prompts to run without one")
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(path, use_fast=tokenizer_mode != "slow", trust_remote_code=False,
                                        local_files_only=True, **kw)
    return tok


def load_generation_config(model: str) -> dict:
    p = os.path.join(model, "generation_config.json")
    if os.path.isfile(p):
        with open(p) as f:
            return json.load(f)
    return {}


class TokenizerGroup:
    """One base tokenizer plus per-adapter tokenizers for LoRA requests."""

    def __init__(self, tokenizer, lora_enabled: bool = False):
        self.tokenizer = tokenizer
        self.lora_enabled = lora_enabled
        self._lora: dict[int, Any] = {}

    def get_lora_tokenizer(self, lora_request: Optional[LoRARequest] = None):
        if lora_request is None or not self.lora_enabled:
            return self.tokenizer
        t = self._lora.get(lora_request.lora_int_id)
        if t is None:
            try:
                t = get_tokenizer(lora_request.lora_path)
            except Exception:
                t = self.tokenizer
            self._lora[lora_request.lora_int_id] = t
        return t

    def encode(self, prompt: str, lora_request=None, add_special_tokens: Optional[bool] = None,
               truncation: Optional[int] = None) -> list[int]:
        tok = self.get_lora_tokenizer(lora_request)
        kw = {}
        if add_special_tokens is not None:
            kw["add_special_tokens"] = add_special_tokens
        ids = tok.encode(prompt, **kw)
        if truncation is not None:
            ids = ids[-truncation:]
        return ids


class Processor:
    def __init__(self, vllm_config: VllmConfig, tokenizer: Optional[TokenizerGroup]):
        self.vllm_config = vllm_config
        self.model_config = vllm_config.model_config
        self.tokenizer = tokenizer
        self.generation_config = load_generation_config(self.model_config.model) \
            if self.model_config.generation_config == "auto" else {}
        self.use_eagle = bool(vllm_config.speculative_config and vllm_config.speculative_config.use_eagle())

    @property
    def eos_token_id(self) -> Optional[int]:
        if self.tokenizer is not None:
            return getattr(self.tokenizer.tokenizer, "eos_token_id", None)
        eos = self.model_config.hf_config.get("eos_token_id")
        return eos[0] if isinstance(eos, list) else eos

    def get_default_sampling_params(self) -> SamplingParams:
        keys = ("temperature", "top_k", "top_p", "min_p", "repetition_penalty")
        kw = {k: self.generation_config[k] for k in keys if self.generation_config.get(k) is not None}
        if "max_new_tokens" in self.generation_config:
            kw["max_tokens"] = self.generation_config["max_new_tokens"]
        return SamplingParams(**kw)

    # ------------------------------------------------------------------
    def _validate_params(self, params: Union[SamplingParams, PoolingParams], lora_request) -> None:
        if isinstance(params, PoolingParams):
            return
        mc = self.model_config
        if params.logprobs is not None and params.logprobs > mc.max_logprobs:
            raise ValueError(f"logprobs={params.logprobs} exceeds max_logprobs={mc.max_logprobs}")
        if params.prompt_logprobs is not None and params.prompt_logprobs > mc.max_logprobs:
            raise ValueError(f"prompt_logprobs={params.prompt_logprobs} exceeds max_logprobs={mc.max_logprobs}")
        V = mc.get_vocab_size()
        if params.allowed_token_ids is not None:
            if not params.allowed_token_ids:
                raise ValueError("allowed_token_ids is empty")
            if any(t < 0 or t >= V for t in params.allowed_token_ids):
                raise ValueError("allowed_token_ids contains a token outside the vocabulary")
        if params.logit_bias and any(t < 0 or t >= V for t in params.logit_bias):
            raise ValueError("logit_bias contains a token outside the vocabulary")
        if params.guided_decoding is not None and self.tokenizer is None:
            raise ValueError("guided decoding needs a tokenizer")
        if lora_request is not None and self.vllm_config.lora_config is None:
            raise ValueError("got a LoRA request but LoRA is not enabled (enable_lora=True)")

    def process_inputs(self, request_id: str, prompt: PromptType, params: Union[SamplingParams, PoolingParams],
                       arrival_time: Optional[float] = None, lora_request: Optional[LoRARequest] = None,
                       tokenization_kwargs: Optional[dict] = None, trace_headers: Optional[dict] = None,
                       priority: int = 0, data_parallel_rank: Optional[int] = None
                       ) -> tuple[Optional[str], EngineCoreRequest]:
        self._validate_params(params, lora_request)
        if data_parallel_rank is not None and not 0 <= data_parallel_rank < \
                self.vllm_config.parallel_config.data_parallel_size:
            raise ValueError(f"data_parallel_rank {data_parallel_rank} out of range")
        arrival_time = time.time() if arrival_time is None else arrival_time
        p = parse_prompt(prompt)
        text = p.get("prompt")
        ids = p.get("prompt_token_ids")
        if ids is None:
            if self.tokenizer is None:
                raise ValueError("text prompts need a tokenizer (skip_tokenizer_init=True was given)")
            trunc = getattr(params, "truncate_prompt_tokens", None)
            ids = self.tokenizer.encode(text, lora_request, truncation=trunc, **(tokenization_kwargs or {}))
        else:
            ids = list(ids)
            trunc = getattr(params, "truncate_prompt_tokens", None)
            if trunc:
                ids = ids[-trunc:]
        if not ids:
            raise ValueError("the prompt is empty")
        V = self.model_config.get_vocab_size()
        mx = max(ids)
        if mx >= V or min(ids) < 0:
            # placeholder tokens of multimodal checkpoints may sit just behind the text vocabulary
            hf = self.model_config.hf_config
            marks = {int(v) for k, v in hf.to_dict().items()
                     if k.endswith(("_token_id", "_token_index")) and isinstance(v, int) and not isinstance(v, bool)}
            if "MolmoForCausalLM" in (self.model_config.architectures or []):
                # its image bracket / patch / column tokens live in 128 embedding rows the head does not have
                V = int(hf.get("embedding_size") or V) + 128
            if min(ids) < 0 or any(t >= V and t not in marks for t in ids):
                raise ValueError(f"token id {mx} is out of vocabulary (size {V})")
        max_len = self.model_config.max_model_len
        if len(ids) >= max_len and isinstance(params, SamplingParams):
            raise ValueError(f"the prompt has {len(ids)} tokens but the model's context is {max_len}; "
                             "shorten the prompt or raise max_model_len")
        if len(ids) > max_len:
            raise ValueError(f"the prompt has {len(ids)} tokens, more than max_model_len={max_len}")

        sampling = pooling = None
        eos = self.eos_token_id
        if isinstance(params, SamplingParams):
            sampling = params.clone()
            if sampling.max_tokens is None:
                sampling.max_tokens = max_len - len(ids)
            sampling.update_from_generation_config(self.generation_config, eos)
            if self.tokenizer is not None:
                sampling.update_from_tokenizer(self.tokenizer.get_lora_tokenizer(lora_request))
        else:
            pooling = params.clone()
        mm = p.get("multi_modal_data")
        if not mm and self.vllm_config.model_config.is_text_encoder_decoder:
            # a plain prompt of a text encoder-decoder is the ENCODER input; the decoder starts from its start token
            mm = {"encoder_token_ids": list(ids)}
            ids = [int(self.vllm_config.model_config.hf_config.get("decoder_start_token_id", 2) or 0)]
            if sampling is not None and p.get("prompt_token_ids") is not None and sampling.max_tokens is not None:
                sampling.max_tokens = min(sampling.max_tokens, max_len - 1)
        mm_inputs = mm_hashes = mm_pos = None
        if mm:
            from ..multimodal import process_multimodal
            ids, mm_inputs, mm_hashes, mm_pos = process_multimodal(self.vllm_config, ids, mm)
        req = EngineCoreRequest(
            request_id=request_id, prompt_token_ids=ids, sampling_params=sampling, eos_token_id=eos,
            arrival_time=arrival_time, lora_request=lora_request, cache_salt=p.get("cache_salt"),
            priority=priority, pooling_params=pooling, mm_inputs=mm_inputs, mm_hashes=mm_hashes,
            mm_placeholders=mm_pos, data_parallel_rank=data_parallel_rank, trace_headers=trace_headers,
            kv_transfer_params=(sampling.extra_args or {}).get("kv_transfer_params") if sampling else None)
        return text, req
