"""Endpoint handlers of the OpenAI-compatible server (reference:
vllm/entrypoints/openai/serving_engine.py, serving_completion.py, serving_chat.py,
serving_embedding.py, serving_tokenization.py, serving_models.py, serving_score.py)."""
from __future__ import annotations

import asyncio
import base64
import json
import time
from http import HTTPStatus
from typing import Any, AsyncGenerator, AsyncIterator, Optional, Union

import numpy as np

from ...engine.core_client import EngineDeadError
from ...inputs import TokensPrompt
from ...logger import init_logger
from ...outputs import RequestOutput
from ...request import LoRARequest
from ...sampling_params import PoolingParams, RequestOutputKind, SamplingParams
from ...tracing import extract_trace_headers
from ...utils import random_uuid
from ..chat_utils import apply_chat_template
from .parsers import get_reasoning_parser, get_tool_parser
from .stream_parsers import (ChatDeltaStream, JsonCallStream, make_reasoning_stream, make_tool_stream,
                             run_tool_stream)
from .protocol import *  # noqa: F401,F403
from .protocol import (ChatCompletionLogProb, ChatCompletionLogProbs, ChatCompletionLogProbsContent,
                       ChatCompletionRequest, ChatCompletionResponse, ChatCompletionResponseChoice,
                       ChatCompletionResponseStreamChoice, ChatCompletionStreamResponse, ChatMessage,
                       CompletionLogProbs, CompletionRequest, CompletionResponse, CompletionResponseChoice,
                       CompletionResponseStreamChoice, CompletionStreamResponse, DeltaFunctionCall, DeltaMessage,
                       DeltaToolCall, DetokenizeRequest, DetokenizeResponse, EmbeddingRequest, EmbeddingResponse,
                       EmbeddingResponseData, ErrorResponse, ModelCard, ModelList, ModelPermission,
                       PromptTokenUsageInfo, ScoreRequest, ScoreResponse, ScoreResponseData, TokenizeRequest,
                       TokenizeResponse, UsageInfo, random_id)

logger = init_logger(__name__)


async def merge_async_iterators(*its: AsyncIterator) -> AsyncGenerator[tuple[int, Any], None]:
    """Yield (index, item) from several async iterators as items arrive."""
    if len(its) == 1:
        async for x in its[0]:
            yield 0, x
        return
    q: asyncio.Queue = asyncio.Queue()
    done = object()

    async def pump(i, it):
        try:
            async for x in it:
                await q.put((i, x))
        except BaseException as e:   # noqa: BLE001 - re-raised in the consumer
            await q.put((i, e))
        finally:
            await q.put((i, done))

    tasks = [asyncio.create_task(pump(i, it)) for i, it in enumerate(its)]
    left = len(its)
    try:
        while left:
            i, x = await q.get()
            if x is done:
                left -= 1
            elif isinstance(x, BaseException):
                raise x
            else:
                yield i, x
    finally:
        for t in tasks:
            t.cancel()


class OpenAIServingModels:
    def __init__(self, engine_client, model_config, served_names: list[str], lora_modules: Optional[list] = None):
        self.engine_client = engine_client
        self.model_config = model_config
        self.served_names = served_names
        self.lora_requests: dict[str, LoRARequest] = {}
        self._next_lora_id = 1
        for m in lora_modules or []:
            self._register(m["name"], m["path"])

    def _register(self, name: str, path: str) -> LoRARequest:
        r = LoRARequest(name, self._next_lora_id, path)
        self._next_lora_id += 1
        self.lora_requests[name] = r
        return r

    def is_base_model(self, name: Optional[str]) -> bool:
        return name is None or name in self.served_names

    def model_name(self, lora: Optional[LoRARequest] = None) -> str:
        return lora.lora_name if lora else self.served_names[0]

    async def show_available_models(self) -> ModelList:
        cards = [ModelCard(id=n, root=self.model_config.model, max_model_len=self.model_config.max_model_len,
                           permission=[ModelPermission()]) for n in self.served_names]
        cards += [ModelCard(id=l.lora_name, root=l.lora_path, parent=self.served_names[0],
                            permission=[ModelPermission()]) for l in self.lora_requests.values()]
        return ModelList(data=cards)

    async def load_lora_adapter(self, name: str, path: str):
        if name in self.lora_requests:
            return ErrorResponse(message=f"adapter {name!r} is already loaded", type="InvalidUserInput", code=400)
        r = self._register(name, path)
        try:
            await self.engine_client.add_lora(r)
        except Exception as e:   # noqa: BLE001
            self.lora_requests.pop(name, None)
            return ErrorResponse(message=str(e), type="BadRequestError", code=400)
        return f"Success: LoRA adapter '{name}' added successfully."

    async def unload_lora_adapter(self, name: str):
        r = self.lora_requests.pop(name, None)
        if r is None:
            return ErrorResponse(message=f"adapter {name!r} is not loaded", type="NotFoundError", code=404)
        await self.engine_client.remove_lora(r.lora_int_id)
        return f"Success: LoRA adapter '{name}' removed successfully."


class OpenAIServing:
    def __init__(self, engine_client, model_config, models: OpenAIServingModels, request_logger=None,
                 return_tokens_as_token_ids: bool = False):
        self.engine_client = engine_client
        self.model_config = model_config
        self.models = models
        self.request_logger = request_logger
        self.return_tokens_as_token_ids = return_tokens_as_token_ids
        self.max_model_len = model_config.max_model_len

    def error(self, message: str, err_type: str = "BadRequestError", status: HTTPStatus = HTTPStatus.BAD_REQUEST):
        return ErrorResponse(message=message, type=err_type, code=status.value)

    def check_model(self, request) -> Optional[ErrorResponse]:
        name = getattr(request, "model", None)
        if self.models.is_base_model(name) or name in self.models.lora_requests:
            return None
        return self.error(f"The model `{name}` does not exist.", "NotFoundError", HTTPStatus.NOT_FOUND)

    def lora_for(self, request) -> Optional[LoRARequest]:
        return self.models.lora_requests.get(getattr(request, "model", None) or "")

    async def tokenizer(self, lora=None):
        return await self.engine_client.get_tokenizer(lora)

    def _tok_str(self, tok, tid: int, as_id: bool) -> str:
        return f"token_id:{tid}" if as_id else tok.decode([tid])

    async def _tokenize_prompt(self, tok, prompt, add_special_tokens: bool, truncate: Optional[int]) -> list[dict]:
        """str | list[str] | list[int] | list[list[int]] -> list of {'prompt', 'prompt_token_ids'}."""
        if isinstance(prompt, str):
            items = [prompt]
        elif prompt and isinstance(prompt[0], int):
            items = [prompt]
        else:
            items = list(prompt or [])
        out = []
        for p in items:
            if isinstance(p, str):
                ids = tok.encode(p, add_special_tokens=add_special_tokens)
                text = p
            else:
                ids, text = list(p), None
            if truncate:
                ids = ids[-truncate:]
            if len(ids) >= self.max_model_len:
                raise ValueError(f"This model's maximum context length is {self.max_model_len} tokens, but the "
                                 f"prompt has {len(ids)} tokens. Please reduce the length of the prompt.")
            out.append({"prompt": text, "prompt_token_ids": ids})
        return out


class OpenAIServingCompletion(OpenAIServing):
    def __init__(self, *a, default_sampling: Optional[dict] = None, **kw):
        super().__init__(*a, **kw)
        self.default_sampling = default_sampling or {}

    async def create_completion(self, request: CompletionRequest, raw_request=None):
        err = self.check_model(request)
        if err:
            return err
        if self.engine_client.errored:
            raise self.engine_client.dead_error
        if request.suffix is not None:
            return self.error("suffix is not supported")
        rid = f"cmpl-{request.request_id or random_uuid()}"
        created = int(time.time())
        lora = self.lora_for(request)
        try:
            tok = await self.tokenizer(lora)
            prompts = await self._tokenize_prompt(tok, request.prompt, request.add_special_tokens,
                                                  request.truncate_prompt_tokens)
        except ValueError as e:
            return self.error(str(e))
        if not prompts:
            return self.error("prompt is required")
        trace = extract_trace_headers(raw_request.headers) if raw_request is not None else None
        gens = []
        try:
            for i, p in enumerate(prompts):
                sp = request.to_sampling_params(self.max_model_len - len(p["prompt_token_ids"]), self.default_sampling)
                gens.append(self.engine_client.generate(TokensPrompt(prompt_token_ids=p["prompt_token_ids"]), sp,
                                                        f"{rid}-{i}", lora_request=lora, trace_headers=trace,
                                                        priority=request.priority))
        except ValueError as e:
            return self.error(str(e))
        merged = merge_async_iterators(*gens)
        model_name = self.models.model_name(lora)
        n_prompts = len(prompts)
        as_ids = bool(request.return_tokens_as_token_ids or self.return_tokens_as_token_ids)
        if request.stream:
            return self._stream(request, merged, rid, created, model_name, n_prompts, prompts, tok, as_ids)
        finals: list[Optional[RequestOutput]] = [None] * n_prompts
        try:
            async for i, res in merged:
                finals[i] = res
        except asyncio.CancelledError:
            return self.error("Client disconnected")
        except ValueError as e:
            return self.error(str(e))
        for i, f in enumerate(finals):
            if f is not None and f.prompt is None:
                f.prompt = prompts[i]["prompt"] if prompts[i]["prompt"] is not None else \
                    tok.decode(prompts[i]["prompt_token_ids"])
        return self._final(finals, request, rid, created, model_name, tok, as_ids)

    def _logprobs(self, token_ids, top_logprobs, num_lp, tok, as_ids, initial_offset=0) -> CompletionLogProbs:
        lp = CompletionLogProbs()
        off = initial_offset
        for i, tid in enumerate(token_ids):
            step = top_logprobs[i] if top_logprobs else None
            if step is None:
                s = self._tok_str(tok, tid, as_ids)
                lp.tokens.append(s); lp.token_logprobs.append(None); lp.top_logprobs.append(None)
            else:
                e = step.get(tid)
                s = self._tok_str(tok, tid, as_ids) if e is None or e.decoded_token is None or as_ids else e.decoded_token
                lp.tokens.append(s)
                lp.token_logprobs.append(max(e.logprob, -9999.0) if e else None)
                lp.top_logprobs.append({(self._tok_str(tok, t, as_ids) if v.decoded_token is None or as_ids
                                         else v.decoded_token): max(v.logprob, -9999.0)
                                        for t, v in list(step.items())[: max(num_lp, 1)]})
            lp.text_offset.append(off)
            off += len(s)
        return lp

    async def _stream(self, request, merged, rid, created, model_name, n_prompts, prompts, tok, as_ids):
        n = request.n
        prev_lens = [0] * (n * n_prompts)
        prompt_tokens = [0] * n_prompts
        completion_tokens = [0] * (n * n_prompts)
        echoed = [False] * (n * n_prompts)
        include_usage = bool(request.stream_options and request.stream_options.include_usage)
        cont_usage = bool(request.stream_options and request.stream_options.continuous_usage_stats)
        try:
            async for pi, res in merged:
                prompt_tokens[pi] = len(res.prompt_token_ids or ())
                for out in res.outputs:
                    i = out.index + pi * n
                    text, ids = out.text, list(out.token_ids)
                    lps = out.logprobs
                    if request.echo and not echoed[i]:
                        echoed[i] = True
                        ptxt = prompts[pi]["prompt"] if prompts[pi]["prompt"] is not None else \
                            tok.decode(prompts[pi]["prompt_token_ids"])
                        text = ptxt + text
                    logprobs = None
                    if request.logprobs is not None and lps is not None:
                        logprobs = self._logprobs(ids, lps, request.logprobs, tok, as_ids, prev_lens[i])
                    prev_lens[i] += len(out.text)
                    completion_tokens[i] += len(out.token_ids)
                    chunk = CompletionStreamResponse(id=rid, created=created, model=model_name, choices=[
                        CompletionResponseStreamChoice(index=i, text=text, logprobs=logprobs,
                                                       finish_reason=out.finish_reason, stop_reason=out.stop_reason)])
                    if include_usage and cont_usage:
                        chunk.usage = UsageInfo(prompt_tokens=prompt_tokens[pi], completion_tokens=completion_tokens[i],
                                                total_tokens=prompt_tokens[pi] + completion_tokens[i])
                    yield f"data: {chunk.model_dump_json(exclude_unset=False)}\n\n"
            if include_usage:
                pt, ct = sum(prompt_tokens), sum(completion_tokens)
                final = CompletionStreamResponse(id=rid, created=created, model=model_name, choices=[],
                                                 usage=UsageInfo(prompt_tokens=pt, completion_tokens=ct,
                                                                 total_tokens=pt + ct))
                yield f"data: {final.model_dump_json(exclude_unset=False, exclude_none=True)}\n\n"
        except (ValueError, EngineDeadError) as e:
            yield f"data: {json.dumps({'error': self.error(str(e)).model_dump()})}\n\n"
        yield "data: [DONE]\n\n"

    def _final(self, finals, request, rid, created, model_name, tok, as_ids) -> CompletionResponse:
        choices = []
        pt = ct = 0
        kv_params = None
        for res in finals:
            ptoks = res.prompt_token_ids or []
            plp = res.prompt_logprobs
            kv_params = res.kv_transfer_params or kv_params
            for out in res.outputs:
                ids, text, lps = list(out.token_ids), out.text, out.logprobs
                if request.echo:
                    if request.max_tokens == 0:
                        ids, lps, text = list(ptoks), plp, res.prompt or ""
                    else:
                        ids = list(ptoks) + ids
                        lps = (plp + lps) if (request.logprobs is not None and plp is not None and lps is not None) \
                            else lps
                        text = (res.prompt or "") + text
                logprobs = None
                if request.logprobs is not None and lps is not None:
                    logprobs = self._logprobs(ids, lps, request.logprobs, tok, as_ids)
                choices.append(CompletionResponseChoice(
                    index=len(choices), text=text, logprobs=logprobs, finish_reason=out.finish_reason,
                    stop_reason=out.stop_reason,
                    prompt_logprobs=_clean_prompt_logprobs(plp) if request.prompt_logprobs is not None else None))
                ct += len(out.token_ids)
            pt += len(ptoks)
        return CompletionResponse(id=rid, created=created, model=model_name, choices=choices,
                                  usage=UsageInfo(prompt_tokens=pt, completion_tokens=ct, total_tokens=pt + ct),
                                  kv_transfer_params=kv_params)


def _clean_prompt_logprobs(plp):
    if plp is None:
        return None
    out = []
    for step in plp:
        if step is None:
            out.append(None)
        else:
            out.append({t: {"logprob": max(v.logprob, -9999.0), "rank": v.rank, "decoded_token": v.decoded_token}
                        for t, v in step.items()})
    return out


class OpenAIServingChat(OpenAIServing):
    def __init__(self, *a, chat_template: Optional[str] = None, tool_parser: Optional[str] = None,
                 reasoning_parser: Optional[str] = None, enable_auto_tools: bool = False,
                 default_sampling: Optional[dict] = None, response_role: str = "assistant", **kw):
        super().__init__(*a, **kw)
        self.chat_template = chat_template
        self.tool_parser = get_tool_parser(tool_parser) if (tool_parser and enable_auto_tools) else None
        self.reasoning_parser = get_reasoning_parser(reasoning_parser) if reasoning_parser else None
        make_tool_stream(self.tool_parser.name) if self.tool_parser is not None else None        # fail at start-up
        make_reasoning_stream(self.reasoning_parser.name) if self.reasoning_parser is not None else None
        self.default_sampling = default_sampling or {}
        self.response_role = response_role

    async def create_chat_completion(self, request: ChatCompletionRequest, raw_request=None):
        err = self.check_model(request)
        if err:
            return err
        if self.engine_client.errored:
            raise self.engine_client.dead_error
        lora = self.lora_for(request)
        try:
            tok = await self.tokenizer(lora)
            tools = [t.model_dump() for t in request.tools] if request.tools and request.tool_choice != "none" else None
            if request.tool_choice == "auto" and self.tool_parser is None:
                return self.error('"auto" tool choice requires --enable-auto-tool-choice and --tool-call-parser')
            text = apply_chat_template(tok, request.messages, chat_template=request.chat_template or self.chat_template,
                                       add_generation_prompt=request.add_generation_prompt,
                                       continue_final_message=request.continue_final_message, tools=tools,
                                       **(request.chat_template_kwargs or {}))
            prompts = await self._tokenize_prompt(tok, text, request.add_special_tokens,
                                                  request.truncate_prompt_tokens)
            p = prompts[0]
            from ..chat_utils import load_media, parse_chat_messages
            _, media = parse_chat_messages(request.messages)
            mm_data = load_media(media) if media else None
            if mm_data:
                hf = self.model_config.hf_config
                subs = []
                for kind, text_marker, names in (("image", "<image>", ("image_token_index", "image_token_id")),
                                                 ("audio", "<audio>", ("audio_token_index", "audio_token_id"))):
                    if kind not in mm_data:
                        continue
                    tid = next((getattr(hf, n) for n in names if getattr(hf, n, None) is not None), None)
                    if tid is None:
                        raise ValueError(f"this model takes no {kind} inputs")
                    subs.append((tok.encode(text_marker, add_special_tokens=False), int(tid)))
                ids, out_ids, i = p["prompt_token_ids"], [], 0
                while i < len(ids):            # the literal placeholder text -> the model's placeholder token
                    for marker, tid in subs:
                        if marker and ids[i:i + len(marker)] == marker:
                            out_ids.append(tid)
                            i += len(marker)
                            break
                    else:
                        out_ids.append(ids[i])
                        i += 1
                p["prompt_token_ids"] = out_ids
            sp = request.to_sampling_params(self.max_model_len - len(p["prompt_token_ids"]), self.default_sampling)
        except Exception as e:   # noqa: BLE001 - template / validation problems are client errors
            logger.debug("chat preprocessing failed", exc_info=True)
            return self.error(str(e))
        rid = f"chatcmpl-{request.request_id or random_uuid()}"
        trace = extract_trace_headers(raw_request.headers) if raw_request is not None else None
        try:
            prompt = TokensPrompt(prompt_token_ids=p["prompt_token_ids"])
            if mm_data:
                prompt["multi_modal_data"] = mm_data
            gen = self.engine_client.generate(prompt, sp, rid, lora_request=lora, trace_headers=trace,
                                              priority=request.priority)
        except ValueError as e:
            return self.error(str(e))
        model_name = self.models.model_name(lora)
        created = int(time.time())
        as_ids = bool(request.return_tokens_as_token_ids or self.return_tokens_as_token_ids)
        if request.stream:
            return self._stream(request, gen, rid, created, model_name, tok, as_ids)
        final = None
        try:
            async for res in gen:
                final = res
        except asyncio.CancelledError:
            return self.error("Client disconnected")
        except ValueError as e:
            return self.error(str(e))
        return self._final(request, final, rid, created, model_name, tok, as_ids)

    def _chat_logprobs(self, ids, lps, top_n, tok, as_ids) -> ChatCompletionLogProbs:
        content = []
        for i, tid in enumerate(ids):
            step = lps[i] if lps else None
            s = self._tok_str(tok, tid, as_ids)
            if step is None or tid not in step:
                content.append(ChatCompletionLogProbsContent(token=s, bytes=list(s.encode("utf-8", "replace"))))
                continue
            e = step[tid]
            tops = []
            for t, v in list(step.items())[: max(top_n or 0, 0)]:
                ts = self._tok_str(tok, t, as_ids) if v.decoded_token is None or as_ids else v.decoded_token
                tops.append(ChatCompletionLogProb(token=ts, logprob=max(v.logprob, -9999.0),
                                                  bytes=list(ts.encode("utf-8", "replace"))))
            content.append(ChatCompletionLogProbsContent(token=s, logprob=max(e.logprob, -9999.0),
                                                         bytes=list(s.encode("utf-8", "replace")), top_logprobs=tops))
        return ChatCompletionLogProbs(content=content)

    def _tool_mode(self, request) -> Optional[str]:
        """How the reply of this request turns into tool calls: "named" (the whole content is the arguments of the
        chosen function), "required" (a JSON array of {name, parameters}, enforced by the grammar), "auto" (the
        model's own call syntax, read by the configured parser) or None."""
        if not request.tools or request.tool_choice in (None, "none"):
            return None
        if isinstance(request.tool_choice, ChatCompletionNamedToolChoiceParam):
            return "named"
        if request.tool_choice == "required":
            return "required"
        return "auto" if self.tool_parser is not None else None

    def _post(self, text: str, request):
        """Whole-text form of the streaming machines: the same calls and argument strings as a streamed reply."""
        from .protocol import FunctionCall, ToolCall
        reasoning = None
        content: Optional[str] = text
        if self.reasoning_parser is not None:
            r, c = make_reasoning_stream(self.reasoning_parser.name).feed(text, final=True)
            reasoning, content = r.strip() or None, c.strip() or None
        calls = []
        mode = self._tool_mode(request)
        if mode == "named":
            calls = [ToolCall(function=FunctionCall(name=request.tool_choice.function.get("name", ""),
                                                    arguments=content or ""))]
            content = None
        elif mode == "required" and content:
            _, found = _run_stream(JsonCallStream(bare=True), content)
            calls = [ToolCall(function=FunctionCall(name=n, arguments=a)) for n, a in found]
            content = None if calls else content
        elif mode == "auto" and content:
            rest, found = run_tool_stream(self.tool_parser.name, content)
            if found:
                calls = [ToolCall(function=FunctionCall(name=n, arguments=a)) for n, a in found]
                content = rest.strip() or None
        return reasoning, content, calls

    def _final(self, request, res: RequestOutput, rid, created, model_name, tok, as_ids) -> ChatCompletionResponse:
        choices = []
        for out in res.outputs:
            reasoning, content, calls = self._post(out.text, request)
            lp = self._chat_logprobs(list(out.token_ids), out.logprobs, request.top_logprobs, tok, as_ids) \
                if request.logprobs and out.logprobs is not None else None
            msg = ChatMessage(role=self.response_role, content=content, reasoning_content=reasoning, tool_calls=calls)
            fr = "tool_calls" if calls and out.finish_reason == "stop" else out.finish_reason
            choices.append(ChatCompletionResponseChoice(index=out.index, message=msg, logprobs=lp, finish_reason=fr,
                                                        stop_reason=out.stop_reason))
        if request.echo:
            last = request.messages[-1].get("content") if request.messages else ""
            if isinstance(last, str):
                for c in choices:
                    c.message.content = last + (c.message.content or "")
        pt = len(res.prompt_token_ids or ())
        ct = sum(len(o.token_ids) for o in res.outputs)
        usage = UsageInfo(prompt_tokens=pt, completion_tokens=ct, total_tokens=pt + ct)
        if res.num_cached_tokens:
            usage.prompt_tokens_details = PromptTokenUsageInfo(cached_tokens=res.num_cached_tokens)
        return ChatCompletionResponse(id=rid, created=created, model=model_name, choices=choices, usage=usage,
                                      prompt_logprobs=_clean_prompt_logprobs(res.prompt_logprobs),
                                      kv_transfer_params=res.kv_transfer_params)

    def _delta_stream(self, request) -> Optional[ChatDeltaStream]:
        """The incremental parser chain of one streamed choice, or None when the text goes out as it is."""
        mode = self._tool_mode(request)
        rname = self.reasoning_parser.name if self.reasoning_parser is not None else None
        if mode is None and rname is None:
            return None
        st = ChatDeltaStream(rname, self.tool_parser.name if mode == "auto" else None)
        if mode == "required":
            st.tool = JsonCallStream(bare=True)
        elif mode == "named":
            st.tool = _NamedCallStream(request.tool_choice.function.get("name", ""))
        return st

    async def _stream(self, request, gen, rid, created, model_name, tok, as_ids):
        n = request.n
        first = [True] * n
        ct = [0] * n
        pt = 0
        include_usage = bool(request.stream_options and request.stream_options.include_usage)
        streams = [self._delta_stream(request) for _ in range(n)]
        tool_ids: list[dict[int, str]] = [{} for _ in range(n)]

        def chunk(i, delta, lp=None, finish=None, stop_reason=None) -> str:
            c = ChatCompletionStreamResponse(id=rid, created=created, model=model_name, choices=[
                ChatCompletionResponseStreamChoice(index=i, delta=delta, logprobs=lp, finish_reason=finish,
                                                   stop_reason=stop_reason)])
            return f"data: {c.model_dump_json(exclude_unset=True)}\n\n"
        try:
            async for res in gen:
                pt = len(res.prompt_token_ids or ())
                for out in res.outputs:
                    i = out.index
                    if first[i]:
                        first[i] = False
                        yield chunk(i, DeltaMessage(role=self.response_role, content=""))
                    ct[i] += len(out.token_ids)
                    lp = self._chat_logprobs(list(out.token_ids), out.logprobs, request.top_logprobs, tok, as_ids) \
                        if request.logprobs and out.logprobs is not None else None
                    finish = out.finish_reason
                    st = streams[i]
                    if st is None:
                        delta = DeltaMessage(content=out.text)
                    else:
                        # reasoning text, the name of a tool call and the fragments of its arguments go out as the
                        # tokens arrive; only what may still be the beginning of a marker is held back
                        d = st.feed(out.text, final=finish is not None)
                        kw = {}
                        if d.content:
                            kw["content"] = d.content
                        if d.reasoning:
                            kw["reasoning_content"] = d.reasoning
                        if d.tools:
                            tcs = []
                            for t in d.tools:
                                fn = DeltaFunctionCall(arguments=t.arguments)
                                if t.name is not None:        # first delta of this call: id, type and name
                                    tool_ids[i][t.index] = ToolCall(function=FunctionCall(name=t.name,
                                                                                          arguments="")).id
                                    fn = DeltaFunctionCall(name=t.name, arguments=t.arguments)
                                    tcs.append(DeltaToolCall(index=t.index, id=tool_ids[i][t.index], type="function",
                                                             function=fn))
                                else:
                                    tcs.append(DeltaToolCall(index=t.index, function=fn))
                            kw["tool_calls"] = tcs
                        if not kw and finish is None and lp is None:
                            continue                         # nothing to tell the client yet
                        delta = DeltaMessage(**kw)
                        if finish == "stop" and st.num_tool_calls:
                            finish = "tool_calls"
                    yield chunk(i, delta, lp, finish, out.stop_reason)
            if include_usage:
                tot = sum(ct)
                c = ChatCompletionStreamResponse(id=rid, created=created, model=model_name, choices=[],
                                                 usage=UsageInfo(prompt_tokens=pt, completion_tokens=tot,
                                                                 total_tokens=pt + tot))
                yield f"data: {c.model_dump_json(exclude_unset=True)}\n\n"
        except (ValueError, EngineDeadError) as e:
            yield f"data: {json.dumps({'error': self.error(str(e)).model_dump()})}\n\n"
        yield "data: [DONE]\n\n"


def _run_stream(st, text: str):
    """One tool-call machine run over a finished text: (content, [(name, arguments)])."""
    content, deltas = st.feed(text, final=True)
    calls: dict[int, list] = {}
    for d in deltas:
        c = calls.setdefault(d.index, ["", ""])
        if d.name is not None:
            c[0] = d.name
        c[1] += d.arguments
    return content, [(c[0], c[1]) for _, c in sorted(calls.items())]


class _NamedCallStream:
    """``tool_choice`` names one function: the generated text (grammar-constrained to its schema) IS the argument
    string of call 0, streamed as it comes."""

    def __init__(self, name: str):
        self.name, self.sent = name, [{"name": False, "args": 0}]

    def feed(self, text: str, final: bool = False):
        from .stream_parsers import ToolDelta
        st = self.sent[0]
        d = ToolDelta(index=0, arguments=text[st["args"]:])
        st["args"] = len(text)
        if not st["name"]:
            st["name"], d.name = True, self.name
        return "", ([d] if (d.name is not None or d.arguments) else [])


class OpenAIServingEmbedding(OpenAIServing):
    async def create_embedding(self, request: EmbeddingRequest, raw_request=None):
        err = self.check_model(request)
        if err:
            return err
        if self.model_config.runner_type != "pooling":
            return self.error("the served model does not produce embeddings (start with --task embed)")
        try:
            tok = await self.tokenizer()
            prompts = await self._tokenize_prompt(tok, request.input, request.add_special_tokens,
                                                  request.truncate_prompt_tokens)
        except ValueError as e:
            return self.error(str(e))
        rid = f"embd-{random_uuid()}"
        gens = [self.engine_client.encode(TokensPrompt(prompt_token_ids=p["prompt_token_ids"]),
                                          request.to_pooling_params(), f"{rid}-{i}", priority=request.priority)
                for i, p in enumerate(prompts)]
        finals = [None] * len(prompts)
        async for i, res in merge_async_iterators(*gens):
            finals[i] = res
        data, ntok = [], 0
        for i, f in enumerate(finals):
            vec = f.outputs.data.float().flatten()
            if request.dimensions:
                vec = vec[: request.dimensions]
                vec = vec / vec.norm().clamp(min=1e-12)
            emb: Union[list[float], str] = vec.tolist()
            if request.encoding_format == "base64":
                emb = base64.b64encode(np.asarray(emb, dtype=np.float32).tobytes()).decode()
            data.append(EmbeddingResponseData(index=i, embedding=emb))
            ntok += len(f.prompt_token_ids)
        return EmbeddingResponse(id=rid, model=self.models.model_name(), data=data,
                                 usage=UsageInfo(prompt_tokens=ntok, total_tokens=ntok))


class OpenAIServingScores(OpenAIServing):
    def __init__(self, *a, embedding: OpenAIServingEmbedding = None, **kw):
        super().__init__(*a, **kw)
        self.embedding = embedding

    async def create_score(self, request: ScoreRequest, raw_request=None):
        import torch
        a = [request.text_1] if isinstance(request.text_1, str) else list(request.text_1)
        b = [request.text_2] if isinstance(request.text_2, str) else list(request.text_2)
        if len(a) == 1:
            a = a * len(b)
        if len(a) != len(b):
            return self.error("score needs 1-to-N or N-to-N inputs")
        if any(x.endswith("ForSequenceClassification") for x in self.model_config.architectures):
            return await self._cross_encoder_scores(request, a, b)
        er = await self.embedding.create_embedding(EmbeddingRequest(model=request.model, input=a + b))
        if isinstance(er, ErrorResponse):
            return er
        vecs = [torch.tensor(d.embedding) for d in er.data]
        data = [ScoreResponseData(index=i, score=float(torch.nn.functional.cosine_similarity(
            vecs[i], vecs[len(a) + i], dim=0))) for i in range(len(a))]
        return ScoreResponse(model=er.model, data=data, usage=er.usage)


async def _cross_encoder_scores(self, request, a: list, b: list):
    """Sequence-classification models: each pair is one prompt (the tokenizer's pair encoding, token type
    ids where the model has them); the class-0 output is the score."""
    from ...sampling_params import PoolingParams
    tok = await self.tokenizer()
    rid = f"score-{random_uuid()}"
    gens, ntok = [], 0
    for i, (x, y) in enumerate(zip(a, b)):
        enc = tok(x, y, truncation=True, max_length=self.model_config.max_model_len)
        ntok += len(enc["input_ids"])
        pp = PoolingParams(token_type_ids=list(enc["token_type_ids"]) if "token_type_ids" in enc else None)
        gens.append(self.engine_client.encode(TokensPrompt(prompt_token_ids=list(enc["input_ids"])), pp,
                                              f"{rid}-{i}"))
    finals = [None] * len(gens)
    async for i, res in merge_async_iterators(*gens):
        finals[i] = res
    data = [ScoreResponseData(index=i, score=float(f.outputs.data.float().flatten()[0]))
            for i, f in enumerate(finals)]
    return ScoreResponse(model=self.models.model_name(), data=data,
                         usage=UsageInfo(prompt_tokens=ntok, total_tokens=ntok))


OpenAIServingScores._cross_encoder_scores = _cross_encoder_scores


class OpenAIServingPooling(OpenAIServing):
    """/pooling (raw pooler output), /classify (class probabilities) and /rerank (query vs documents).
    Reference: vllm/entrypoints/openai/serving_pooling.py, serving_classification.py, serving_score.py."""

    def __init__(self, *a, embedding: OpenAIServingEmbedding = None, scores: "OpenAIServingScores" = None, **kw):
        super().__init__(*a, **kw)
        self.embedding, self.scores = embedding, scores

    async def _pool(self, texts, params, rid, priority=0, truncate=None):
        tok = await self.tokenizer()
        prompts = await self._tokenize_prompt(tok, texts, True, truncate)
        gens = [self.engine_client.encode(TokensPrompt(prompt_token_ids=p["prompt_token_ids"]), params, f"{rid}-{i}",
                                          priority=priority) for i, p in enumerate(prompts)]
        finals = [None] * len(prompts)
        async for i, res in merge_async_iterators(*gens):
            finals[i] = res
        return finals

    async def create_pooling(self, request, raw_request=None):
        from .protocol import PoolingResponse, PoolingResponseData
        err = self.check_model(request)
        if err:
            return err
        if self.model_config.runner_type != "pooling":
            return self.error("the served model is not a pooling model")
        rid = f"pool-{random_uuid()}"
        try:
            finals = await self._pool(request.input, request.to_pooling_params(), rid, request.priority,
                                      request.truncate_prompt_tokens)
        except ValueError as e:
            return self.error(str(e))
        ntok = sum(len(f.prompt_token_ids) for f in finals)
        data = [PoolingResponseData(index=i, data=f.outputs.data.float().tolist()) for i, f in enumerate(finals)]
        return PoolingResponse(id=rid, model=self.models.model_name(), data=data,
                               usage=UsageInfo(prompt_tokens=ntok, total_tokens=ntok))

    async def create_classify(self, request, raw_request=None):
        import torch
        from ...sampling_params import PoolingParams
        from .protocol import ClassificationData, ClassificationResponse
        err = self.check_model(request)
        if err:
            return err
        if self.model_config.runner_type != "pooling":
            return self.error("the served model is not a classification model (start with --task classify)")
        rid = f"classify-{random_uuid()}"
        texts = [request.input] if isinstance(request.input, str) else list(request.input)
        try:
            finals = await self._pool(texts, PoolingParams(task="classify"), rid, request.priority,
                                      request.truncate_prompt_tokens)
        except ValueError as e:
            return self.error(str(e))
        id2label = getattr(self.model_config.hf_config, "id2label", None) or {}
        data, ntok = [], 0
        for i, f in enumerate(finals):
            probs = f.outputs.data.float().flatten()
            if probs.numel() > 1 and not torch.isclose(probs.sum(), torch.tensor(1.0), atol=1e-3):
                probs = torch.softmax(probs, dim=-1)
            k = int(probs.argmax())
            label = id2label.get(k, id2label.get(str(k))) if hasattr(id2label, "get") else None
            data.append(ClassificationData(index=i, label=label, probs=probs.tolist(), num_classes=probs.numel()))
            ntok += len(f.prompt_token_ids)
        return ClassificationResponse(id=rid, model=self.models.model_name(), data=data,
                                      usage=UsageInfo(prompt_tokens=ntok, total_tokens=ntok))

    async def create_rerank(self, request, raw_request=None):
        from .protocol import RerankDocument, RerankResponse, RerankResult, RerankUsage
        sr = await self.scores.create_score(ScoreRequest(model=request.model, text_1=request.query,
                                                         text_2=list(request.documents)))
        if isinstance(sr, ErrorResponse):
            return sr
        rows = sorted(sr.data, key=lambda d: -d.score)
        if request.top_n and request.top_n > 0:
            rows = rows[: request.top_n]
        return RerankResponse(model=sr.model, usage=RerankUsage(total_tokens=sr.usage.total_tokens),
                              results=[RerankResult(index=d.index, document=RerankDocument(text=request.documents[d.index]),
                                                    relevance_score=d.score) for d in rows])


class OpenAIServingResponses(OpenAIServing):
    """OpenAI Responses API on top of the chat pipeline: ``input`` (text or message items) and
    ``instructions`` become chat messages; the reply is returned as response output items, with the
    reasoning parser's output as a reasoning item.  Stored responses can be fetched, cancelled and chained
    through ``previous_response_id``.  Reference: vllm/entrypoints/openai/serving_responses.py."""

    def __init__(self, *a, chat: "OpenAIServingChat" = None, **kw):
        super().__init__(*a, **kw)
        self.chat = chat
        self.store: dict[str, tuple[Any, list[dict]]] = {}

    @staticmethod
    def _messages(request, history: list[dict]) -> list[dict]:
        msgs = list(history)
        if request.instructions and not history:
            msgs.append({"role": "system", "content": request.instructions})
        if isinstance(request.input, str):
            msgs.append({"role": "user", "content": request.input})
        else:
            for item in request.input:
                content = item.get("content", "")
                if isinstance(content, list):       # input_text / output_text parts
                    content = "".join(p.get("text", "") for p in content if isinstance(p, dict))
                msgs.append({"role": item.get("role", "user"), "content": content})
        return msgs

    async def create_responses(self, request, raw_request=None):
        from .protocol import (ResponseOutputMessage, ResponseOutputText, ResponseReasoningItem, ResponsesResponse,
                               ResponseUsage)
        if request.stream:
            return self.error("streaming responses are not supported; use /v1/chat/completions for streaming")
        history: list[dict] = []
        if request.previous_response_id:
            prev = self.store.get(request.previous_response_id)
            if prev is None:
                return self.error(f"response {request.previous_response_id} not found", "NotFoundError",
                                  HTTPStatus.NOT_FOUND)
            history = prev[1]
        msgs = self._messages(request, history)
        kw = {"model": request.model, "messages": msgs, "priority": request.priority}
        for src, dst in (("max_output_tokens", "max_completion_tokens"), ("temperature", "temperature"),
                         ("top_p", "top_p")):
            if getattr(request, src) is not None:
                kw[dst] = getattr(request, src)
        res = await self.chat.create_chat_completion(ChatCompletionRequest(**kw), raw_request)
        if isinstance(res, ErrorResponse):
            return res
        msg = res.choices[0].message
        out: list[Any] = []
        if getattr(msg, "reasoning_content", None):
            out.append(ResponseReasoningItem(content=[{"type": "reasoning_text", "text": msg.reasoning_content}]))
        out.append(ResponseOutputMessage(content=[ResponseOutputText(text=msg.content or "")]))
        length = res.choices[0].finish_reason == "length"
        resp = ResponsesResponse(
            model=res.model, output=out, instructions=request.instructions, metadata=request.metadata,
            temperature=request.temperature, top_p=request.top_p, max_output_tokens=request.max_output_tokens,
            previous_response_id=request.previous_response_id, status="incomplete" if length else "completed",
            incomplete_details={"reason": "max_output_tokens"} if length else None,
            usage=ResponseUsage(input_tokens=res.usage.prompt_tokens, output_tokens=res.usage.completion_tokens or 0,
                                total_tokens=res.usage.total_tokens))
        if request.store:
            self.store[resp.id] = (resp, msgs + [{"role": "assistant", "content": msg.content or ""}])
            while len(self.store) > 1024:
                self.store.pop(next(iter(self.store)))
        return resp

    async def retrieve_responses(self, response_id: str):
        hit = self.store.get(response_id)
        if hit is None:
            return self.error(f"response {response_id} not found", "NotFoundError", HTTPStatus.NOT_FOUND)
        return hit[0]

    async def cancel_responses(self, response_id: str):
        hit = self.store.get(response_id)
        if hit is None:
            return self.error(f"response {response_id} not found", "NotFoundError", HTTPStatus.NOT_FOUND)
        return hit[0]          # responses run in the foreground here: nothing is left to cancel


def parse_multipart_form(content_type: str, body: bytes) -> dict:
    """multipart/form-data -> {field: str | (filename, bytes)} with the standard library's MIME parser
    (no multipart package is installed here)."""
    from email.parser import BytesParser
    from email.policy import HTTP
    if "multipart/form-data" not in (content_type or ""):
        raise ValueError("expected multipart/form-data")
    msg = BytesParser(policy=HTTP).parsebytes(b"Content-Type: " + content_type.encode() + b"\r\n\r\n" + body)
    out: dict = {}
    for part in msg.iter_parts():
        name = part.get_param("name", header="content-disposition")
        if name is None:
            continue
        fname = part.get_filename()
        data = part.get_payload(decode=True) or b""
        out[name] = (fname, data) if fname is not None else data.decode("utf-8", "replace")
    return out


class OpenAIServingTranscription(OpenAIServing):
    """/v1/audio/transcriptions and /v1/audio/translations for encoder-decoder speech models (Whisper): the
    uploaded clip is cut into windows of the encoder's length, every window is decoded from the task prompt
    ``<|startoftranscript|><|lang|><|transcribe or translate|><|notimestamps|>`` and the texts are joined.
    Reference: vllm/entrypoints/openai/serving_transcription.py, speech_to_text.py."""
    WINDOW_S = 30

    async def create(self, raw_request, task: str = "transcribe"):
        from ..chat_utils import _decode_wav
        if self.engine_client.errored:
            raise self.engine_client.dead_error
        try:
            form = parse_multipart_form(raw_request.headers.get("content-type", ""), await raw_request.body())
            if "file" not in form or isinstance(form["file"], str):
                raise ValueError("the form needs a 'file' part with the audio (WAV)")
            wave, rate = _decode_wav(form["file"][1])
            lang = (form.get("language") or "en").strip().lower()
            fmt = (form.get("response_format") or "json").strip().lower()
            if fmt not in ("json", "text", "verbose_json"):
                raise ValueError("response_format must be json, text or verbose_json")
            temperature = float(form.get("temperature") or 0.0)
            tok = await self.tokenizer(None)
            vocab = tok.get_vocab()
            names = ["<|startoftranscript|>", f"<|{lang}|>", f"<|{task}|>", "<|notimestamps|>"]
            missing = [n for n in names if n not in vocab]
            if missing:
                raise ValueError(f"the tokenizer lacks {missing}: not a speech-to-text model, or an unknown language")
            prefix = [vocab[n] for n in names]
            hint = form.get("prompt")
            if hint and "<|startofprev|>" in vocab:
                prefix = [vocab["<|startofprev|>"]] + tok.encode(" " + hint.strip(), add_special_tokens=False) + prefix
        except Exception as e:   # noqa: BLE001 - malformed uploads are client errors
            return self.error(str(e))
        win = self.WINDOW_S * rate
        pieces = [wave[i:i + win] for i in range(0, max(len(wave), 1), win)] or [wave]
        rid = f"{'transcribe' if task == 'transcribe' else 'translate'}-{random_uuid()}"
        texts = []
        for k, piece in enumerate(pieces):
            sp = SamplingParams(temperature=temperature, max_tokens=max(self.max_model_len - len(prefix), 1))
            prompt = TokensPrompt(prompt_token_ids=list(prefix), multi_modal_data={"audio": (piece, rate)})
            final = None
            try:
                async for out in self.engine_client.generate(prompt, sp, f"{rid}-{k}"):
                    final = out
            except ValueError as e:
                return self.error(str(e))
            texts.append(final.outputs[0].text.strip() if final is not None else "")
        text = " ".join(t for t in texts if t)
        if fmt == "text":
            return text
        body = {"text": text}
        if fmt == "verbose_json":
            body.update(task=task, language=lang, duration=round(len(wave) / float(rate), 3))
        return body


class OpenAIServingTokenization(OpenAIServing):
    def __init__(self, *a, chat_template: Optional[str] = None, **kw):
        super().__init__(*a, **kw)
        self.chat_template = chat_template

    async def create_tokenize(self, request: TokenizeRequest, raw_request=None):
        err = self.check_model(request)
        if err:
            return err
        tok = await self.tokenizer(self.lora_for(request))
        try:
            if request.messages is not None:
                text = apply_chat_template(tok, request.messages, chat_template=self.chat_template,
                                           add_generation_prompt=request.add_generation_prompt)
                ids = tok.encode(text, add_special_tokens=False)
            elif request.prompt is not None:
                ids = tok.encode(request.prompt, add_special_tokens=request.add_special_tokens)
            else:
                return self.error("either prompt or messages is required")
        except Exception as e:   # noqa: BLE001
            return self.error(str(e))
        strs = tok.convert_ids_to_tokens(ids) if request.return_token_strs else None
        return TokenizeResponse(count=len(ids), max_model_len=self.max_model_len, tokens=list(ids), token_strs=strs)

    async def create_detokenize(self, request: DetokenizeRequest, raw_request=None):
        err = self.check_model(request)
        if err:
            return err
        tok = await self.tokenizer(self.lora_for(request))
        return DetokenizeResponse(prompt=tok.decode(request.tokens))
