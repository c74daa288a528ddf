"""Offline inference API (reference: vllm/entrypoints/llm.py:64-1745)."""
from __future__ import annotations

import itertools
from typing import Any, Callable, Optional, Sequence, Union

from ..engine.arg_utils import EngineArgs
from ..engine.llm_engine import LLMEngine
from ..inputs import PromptType, TokensPrompt
from ..logger import init_logger
from ..outputs import (BeamSearchOutput, BeamSearchSequence, ClassificationRequestOutput, EmbeddingRequestOutput,
                       PoolingRequestOutput, RequestOutput, ScoringOutput, ScoringRequestOutput)
from ..request import LoRARequest
from ..sampling_params import BeamSearchParams, PoolingParams, RequestOutputKind, SamplingParams
from ..utils import Counter

logger = init_logger(__name__)


class LLM:
    def __init__(self, model: str, tokenizer: Optional[str] = None, tokenizer_mode: str = "auto",
                 skip_tokenizer_init: bool = False, trust_remote_code: bool = False,
                 tensor_parallel_size: int = 1, dtype: str = "auto", quantization: Optional[str] = None,
                 seed: int = 0, gpu_memory_utilization: float = 0.9, swap_space: float = 4,
                 enforce_eager: bool = False, **kwargs):
        kwargs.setdefault("disable_log_stats", True)
        engine_args = EngineArgs(model=model, tokenizer=tokenizer, tokenizer_mode=tokenizer_mode,
                                 skip_tokenizer_init=skip_tokenizer_init, trust_remote_code=trust_remote_code,
                                 tensor_parallel_size=tensor_parallel_size, dtype=dtype, quantization=quantization,
                                 seed=seed, gpu_memory_utilization=gpu_memory_utilization, swap_space=swap_space,
                                 enforce_eager=enforce_eager, **kwargs)
        self.llm_engine = LLMEngine.from_engine_args(engine_args)
        self.request_counter = Counter()
        self.default_sampling_params: Optional[SamplingParams] = None

    @classmethod
    def from_engine_args(cls, engine_args: EngineArgs) -> "LLM":
        self = cls.__new__(cls)
        self.llm_engine = LLMEngine.from_engine_args(engine_args)
        self.request_counter = Counter()
        self.default_sampling_params = None
        return self

    def get_tokenizer(self, lora_request: Optional[LoRARequest] = None):
        return self.llm_engine.get_tokenizer(lora_request)

    def get_default_sampling_params(self) -> SamplingParams:
        if self.default_sampling_params is None:
            self.default_sampling_params = self.llm_engine.processor.get_default_sampling_params()
        return self.default_sampling_params.clone()

    # ------------------------------------------------------------------
    def generate(self, prompts: Union[PromptType, Sequence[PromptType], None] = None,
                 sampling_params: Union[SamplingParams, Sequence[SamplingParams], None] = None,
                 prompt_token_ids: Union[list[int], list[list[int]], None] = None, use_tqdm: bool = False,
                 lora_request: Union[LoRARequest, list[LoRARequest], None] = None,
                 priority: Optional[list[int]] = None) -> list[RequestOutput]:
        if self.llm_engine.model_config.runner_type != "generate":
            raise ValueError("LLM.generate() needs a generative model; use encode()/embed() for pooling models")
        prompts = self._normalise_prompts(prompts, prompt_token_ids)
        if sampling_params is None:
            sampling_params = self.get_default_sampling_params()
        self._add_requests(prompts, sampling_params, lora_request, priority)
        return self._run_engine(use_tqdm)

    @staticmethod
    def _normalise_prompts(prompts, prompt_token_ids) -> list[PromptType]:
        if prompt_token_ids is not None:
            if prompts is not None and not isinstance(prompts, (str, dict)) and len(prompts) != len(prompt_token_ids):
                raise ValueError("prompts and prompt_token_ids must have the same length")
            if prompt_token_ids and isinstance(prompt_token_ids[0], int):
                prompt_token_ids = [prompt_token_ids]
            return [TokensPrompt(prompt_token_ids=list(ids)) for ids in prompt_token_ids]
        if prompts is None:
            raise ValueError("either prompts or prompt_token_ids is required")
        if isinstance(prompts, (str, dict)):
            return [prompts]
        if isinstance(prompts, (list, tuple)) and prompts and isinstance(prompts[0], int):
            return [TokensPrompt(prompt_token_ids=list(prompts))]
        return list(prompts)

    def _add_requests(self, prompts, params, lora_request=None, priority=None) -> list[str]:
        n = len(prompts)
        if isinstance(params, (list, tuple)) and len(params) != n:
            raise ValueError("the number of params must match the number of prompts")
        if isinstance(lora_request, list) and len(lora_request) != n:
            raise ValueError("the number of lora_requests must match the number of prompts")
        ids = []
        for i, p in enumerate(prompts):
            sp = params[i] if isinstance(params, (list, tuple)) else params
            if isinstance(sp, SamplingParams):
                sp = sp.clone()
                sp.output_kind = RequestOutputKind.FINAL_ONLY
            lr = lora_request[i] if isinstance(lora_request, list) else lora_request
            rid = str(next(self.request_counter))
            self.llm_engine.add_request(rid, p, sp, lora_request=lr, priority=priority[i] if priority else 0)
            ids.append(rid)
        return ids

    def _run_engine(self, use_tqdm: bool = False) -> list:
        pbar = None
        if use_tqdm:
            from tqdm import tqdm
            pbar = tqdm(total=self.llm_engine.get_num_unfinished_requests(), desc="Processed prompts")
        outputs = []
        while self.llm_engine.has_unfinished_requests():
            for out in self.llm_engine.step():
                if out.finished:
                    outputs.append(out)
                    if pbar is not None:
                        pbar.update(1)
        if pbar is not None:
            pbar.close()
        return sorted(outputs, key=lambda o: int(o.request_id))

    # ---- chat ------------------------------------------------------------------------
    def chat(self, messages, sampling_params=None, use_tqdm: bool = False, lora_request=None,
             chat_template: Optional[str] = None, add_generation_prompt: bool = True,
             continue_final_message: bool = False, tools: Optional[list[dict]] = None,
             chat_template_kwargs: Optional[dict] = None) -> list[RequestOutput]:
        from .chat_utils import apply_chat_template
        convs = messages if messages and isinstance(messages[0], list) else [messages]
        tok = self.get_tokenizer(lora_request if not isinstance(lora_request, list) else None)
        prompts = [apply_chat_template(tok, c, chat_template=chat_template,
                                       add_generation_prompt=add_generation_prompt,
                                       continue_final_message=continue_final_message, tools=tools,
                                       **(chat_template_kwargs or {})) for c in convs]
        return self.generate(prompts, sampling_params, use_tqdm=use_tqdm, lora_request=lora_request)

    # ---- beam search -------------------------------------------------------------------
    def beam_search(self, prompts: list[PromptType], params: BeamSearchParams) -> list[BeamSearchOutput]:
        """Width-limited best-first expansion, one engine call per step with logprobs=2*width."""
        tok = None if self.llm_engine.tokenizer is None else self.get_tokenizer()
        eos = self.llm_engine.processor.eos_token_id
        W = params.beam_width

        def score(b: BeamSearchSequence, plen: int) -> float:
            n = len(b.tokens) - plen
            return b.cum_logprob / (max(n, 1) ** params.length_penalty)

        results = []
        for p in self._normalise_prompts(prompts, None):
            _, req = self.llm_engine.processor.process_inputs("beam", p, SamplingParams(max_tokens=1))
            plen = len(req.prompt_token_ids)
            beams = [BeamSearchSequence(tokens=list(req.prompt_token_ids), logprobs=[])]
            done: list[BeamSearchSequence] = []
            sp = SamplingParams(logprobs=2 * W, max_tokens=1, temperature=params.temperature)
            for _ in range(params.max_tokens):
                if not beams:
                    break
                outs = self.generate([TokensPrompt(prompt_token_ids=b.tokens) for b in beams], sp)
                cand = []
                for b, o in zip(beams, outs):
                    lp = o.outputs[0].logprobs[0]
                    for tid, l in lp.items():
                        nb = BeamSearchSequence(tokens=b.tokens + [tid], logprobs=b.logprobs + [lp],
                                                cum_logprob=b.cum_logprob + l.logprob)
                        if tid == eos and not params.ignore_eos:
                            nb.finish_reason, nb.stop_reason = "stop", eos
                            done.append(nb)
                        else:
                            cand.append(nb)
                beams = sorted(cand, key=lambda x: score(x, plen), reverse=True)[:W]
            done.extend(beams)
            best = sorted(done, key=lambda x: score(x, plen), reverse=True)[:W]
            for b in best:
                if tok is not None:
                    b.text = tok.decode(b.tokens[plen:])
            results.append(BeamSearchOutput(best))
        return results

    # ---- pooling -------------------------------------------------------------------------
    def encode(self, prompts, pooling_params: Optional[PoolingParams] = None, use_tqdm: bool = False,
               lora_request=None, prompt_token_ids=None) -> list[PoolingRequestOutput]:
        if self.llm_engine.model_config.runner_type != "pooling":
            raise ValueError("LLM.encode() needs a pooling model (task=embed/classify/score)")
        prompts = self._normalise_prompts(prompts, prompt_token_ids)
        self._add_requests(prompts, pooling_params or PoolingParams(), lora_request)
        return self._run_engine(use_tqdm)

    def embed(self, prompts, pooling_params=None, **kw) -> list[EmbeddingRequestOutput]:
        return [EmbeddingRequestOutput.from_base(o) for o in self.encode(prompts, pooling_params, **kw)]

    def classify(self, prompts, **kw) -> list[ClassificationRequestOutput]:
        return [ClassificationRequestOutput.from_base(o) for o in self.encode(prompts, **kw)]

    def score(self, text_1, text_2, **kw) -> list[ScoringRequestOutput]:
        """Cosine similarity of embeddings (1-to-1, 1-to-N or N-to-N)."""
        import torch
        a = [text_1] if isinstance(text_1, (str, dict)) else list(text_1)
        b = [text_2] if isinstance(text_2, (str, dict)) else list(text_2)
        if len(a) == 1:
            a = a * len(b)
        if len(a) != len(b):
            raise ValueError("score() needs 1-to-N or N-to-N inputs")
        mc = self.llm_engine.model_config
        if any(x.endswith("ForSequenceClassification") for x in mc.architectures):
            return self._cross_encoder_score(a, b, **kw)
        embs = self.encode(a + b, **kw)
        out = []
        for i in range(len(a)):
            x, y = embs[i].outputs.data.float().flatten(), embs[len(a) + i].outputs.data.float().flatten()
            s = torch.nn.functional.cosine_similarity(x, y, dim=0).item()
            out.append(ScoringRequestOutput(f"{embs[i].request_id}_{embs[len(a) + i].request_id}", ScoringOutput(s),
                                            embs[i].prompt_token_ids + embs[len(a) + i].prompt_token_ids, True))
        return out

    def _cross_encoder_score(self, a: list, b: list, **kw) -> list[ScoringRequestOutput]:
        """Sequence-classification models score a pair in one pass: both texts in one prompt, joined the
        way the tokenizer joins pairs, the second text marked by token type ids where the model has them.
        Reference: vllm/entrypoints/llm.py (_cross_encoding_score)."""
        tok = self.get_tokenizer()
        if tok is None:
            raise ValueError("scoring text pairs with a cross-encoder needs the tokenizer")
        out = []
        for x, y in zip(a, b):
            enc = tok(x, y, truncation=True, max_length=self.llm_engine.model_config.max_model_len)
            pp = PoolingParams(token_type_ids=list(enc["token_type_ids"]) if "token_type_ids" in enc else None)
            o = self.encode(None, pooling_params=pp, prompt_token_ids=[list(enc["input_ids"])], **kw)[0]
            out.append(ScoringRequestOutput(o.request_id, ScoringOutput(float(o.outputs.data.float().flatten()[0])),
                                            o.prompt_token_ids, True))
        return out

    # ---- engine utilities -------------------------------------------------------------------
    def collective_rpc(self, method: Union[str, Callable], timeout: Optional[float] = None, args: tuple = (),
                       kwargs: Optional[dict] = None) -> list[Any]:
        return self.llm_engine.collective_rpc(method, timeout, args, kwargs)

    def apply_model(self, func: Callable) -> list:
        return self.collective_rpc(lambda w: func(w.get_model()))

    def start_profile(self): self.llm_engine.start_profile()
    def stop_profile(self): self.llm_engine.stop_profile()
    def reset_prefix_cache(self, device=None) -> bool: return self.llm_engine.reset_prefix_cache(device)
    def sleep(self, level: int = 1): self.reset_prefix_cache(); self.llm_engine.sleep(level)
    def wake_up(self, tags: Optional[list[str]] = None): self.llm_engine.wake_up(tags)
    def get_metrics(self) -> list: return self.llm_engine.get_metrics()

    def shutdown(self):
        self.llm_engine.shutdown()
