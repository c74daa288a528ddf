"""OpenAI-compatible request / response schemas (reference:
vllm/entrypoints/openai/protocol.py; the same field names and defaults, including
the engine-specific sampling extras such as top_k, min_p, repetition_penalty,
ignore_eos, guided_* and kv_transfer_params)."""
from __future__ import annotations

import time
import uuid
from typing import Any, Literal, Optional, Union

from pydantic import BaseModel, ConfigDict, Field, model_validator

from ...sampling_params import (BeamSearchParams, GuidedDecodingParams, PoolingParams, RequestOutputKind,
                                SamplingParams)


def random_id(prefix: str) -> str:
    return f"{prefix}-{uuid.uuid4().hex}"


class OpenAIBaseModel(BaseModel):
    model_config = ConfigDict(extra="allow")


class ErrorResponse(OpenAIBaseModel):
    object: str = "error"
    message: str
    type: str
    param: Optional[str] = None
    code: int


class ModelPermission(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("modelperm"))
    object: str = "model_permission"
    created: int = Field(default_factory=lambda: int(time.time()))
    allow_create_engine: bool = False
    allow_sampling: bool = True
    allow_logprobs: bool = True
    allow_search_indices: bool = False
    allow_view: bool = True
    allow_fine_tuning: bool = False
    organization: str = "*"
    group: Optional[str] = None
    is_blocking: bool = False


class ModelCard(OpenAIBaseModel):
    id: str
    object: str = "model"
    created: int = Field(default_factory=lambda: int(time.time()))
    owned_by: str = "vda"
    root: Optional[str] = None
    parent: Optional[str] = None
    max_model_len: Optional[int] = None
    permission: list[ModelPermission] = Field(default_factory=list)


class ModelList(OpenAIBaseModel):
    object: str = "list"
    data: list[ModelCard] = Field(default_factory=list)


class PromptTokenUsageInfo(OpenAIBaseModel):
    cached_tokens: Optional[int] = None


class UsageInfo(OpenAIBaseModel):
    prompt_tokens: int = 0
    total_tokens: int = 0
    completion_tokens: Optional[int] = 0
    prompt_tokens_details: Optional[PromptTokenUsageInfo] = None


class StreamOptions(OpenAIBaseModel):
    include_usage: Optional[bool] = True
    continuous_usage_stats: Optional[bool] = False


class ResponseFormat(OpenAIBaseModel):
    type: Literal["text", "json_object", "json_schema", "structural_tag"]
    json_schema: Optional[dict] = None
    structures: Optional[list[dict]] = None          # structural_tag: [{"begin", "schema", "end"}]
    triggers: Optional[list[str]] = None


class _SamplingMixin(OpenAIBaseModel):
    # standard
    frequency_penalty: Optional[float] = 0.0
    presence_penalty: Optional[float] = 0.0
    logit_bias: Optional[dict[str, float]] = None
    max_tokens: Optional[int] = None
    n: int = 1
    seed: Optional[int] = None
    stop: Optional[Union[str, list[str]]] = None
    stream: Optional[bool] = False
    stream_options: Optional[StreamOptions] = None
    temperature: Optional[float] = None
    top_p: Optional[float] = None
    user: Optional[str] = None
    # engine extras
    best_of: Optional[int] = None
    use_beam_search: bool = False
    top_k: Optional[int] = None
    min_p: Optional[float] = None
    repetition_penalty: Optional[float] = None
    length_penalty: float = 1.0
    stop_token_ids: Optional[list[int]] = None
    include_stop_str_in_output: bool = False
    ignore_eos: bool = False
    min_tokens: int = 0
    skip_special_tokens: bool = True
    spaces_between_special_tokens: bool = True
    truncate_prompt_tokens: Optional[int] = None
    allowed_token_ids: Optional[list[int]] = None
    bad_words: Optional[list[str]] = None
    prompt_logprobs: Optional[int] = None
    response_format: Optional[ResponseFormat] = None
    guided_json: Optional[Union[str, dict]] = None
    guided_regex: Optional[str] = None
    guided_choice: Optional[list[str]] = None
    guided_grammar: Optional[str] = None
    guided_decoding_backend: Optional[str] = None
    guided_whitespace_pattern: Optional[str] = None
    priority: int = 0
    request_id: Optional[str] = None
    kv_transfer_params: Optional[dict[str, Any]] = None
    vllm_xargs: Optional[dict[str, Any]] = None

    def _guided(self) -> Optional[GuidedDecodingParams]:
        js = self.guided_json
        json_object = None
        rf = self.response_format
        if rf is not None:
            if rf.type == "structural_tag":
                return GuidedDecodingParams(structural_tag={"structures": rf.structures, "triggers": rf.triggers},
                                            backend=self.guided_decoding_backend)
            if rf.type == "json_object":
                json_object = True
            elif rf.type == "json_schema" and rf.json_schema:
                js = rf.json_schema.get("schema", rf.json_schema)
        if not any((js, self.guided_regex, self.guided_choice, self.guided_grammar, json_object)):
            return None
        return GuidedDecodingParams(json=js, regex=self.guided_regex, choice=self.guided_choice,
                                    grammar=self.guided_grammar, json_object=json_object,
                                    backend=self.guided_decoding_backend,
                                    whitespace_pattern=self.guided_whitespace_pattern)

    def _common_sampling(self, default_max_tokens: int, logprobs: Optional[int], defaults: Optional[dict]) -> SamplingParams:
        d = defaults or {}
        max_tokens = self.max_tokens if self.max_tokens is not None else d.get("max_tokens", default_max_tokens)
        max_tokens = min(max_tokens, default_max_tokens)
        extra = dict(self.vllm_xargs or {})
        if self.kv_transfer_params:
            extra["kv_transfer_params"] = self.kv_transfer_params
        return SamplingParams(
            n=self.n, best_of=self.best_of, presence_penalty=self.presence_penalty or 0.0,
            frequency_penalty=self.frequency_penalty or 0.0,
            repetition_penalty=self.repetition_penalty if self.repetition_penalty is not None
            else d.get("repetition_penalty", 1.0),
            temperature=self.temperature if self.temperature is not None else d.get("temperature", 1.0),
            top_p=self.top_p if self.top_p is not None else d.get("top_p", 1.0),
            top_k=self.top_k if self.top_k is not None else d.get("top_k", 0),
            min_p=self.min_p if self.min_p is not None else d.get("min_p", 0.0),
            seed=self.seed, stop=self.stop, stop_token_ids=self.stop_token_ids, bad_words=self.bad_words,
            include_stop_str_in_output=self.include_stop_str_in_output, ignore_eos=self.ignore_eos,
            max_tokens=max_tokens, min_tokens=self.min_tokens, logprobs=logprobs,
            prompt_logprobs=self.prompt_logprobs, skip_special_tokens=self.skip_special_tokens,
            spaces_between_special_tokens=self.spaces_between_special_tokens,
            truncate_prompt_tokens=self.truncate_prompt_tokens,
            logit_bias={int(k): v for k, v in self.logit_bias.items()} if self.logit_bias else None,
            allowed_token_ids=self.allowed_token_ids, guided_decoding=self._guided(),
            output_kind=RequestOutputKind.DELTA if self.stream else RequestOutputKind.FINAL_ONLY,
            extra_args=extra or None)

    def to_beam_search_params(self, default_max_tokens: int) -> BeamSearchParams:
        return BeamSearchParams(beam_width=self.n, max_tokens=self.max_tokens or default_max_tokens,
                                ignore_eos=self.ignore_eos, temperature=self.temperature or 0.0,
                                length_penalty=self.length_penalty,
                                include_stop_str_in_output=self.include_stop_str_in_output)

    @model_validator(mode="after")
    def _check(self):
        if self.stream_options and not self.stream:
            raise ValueError("stream_options can only be set when stream=true")
        if self.prompt_logprobs is not None and self.stream:
            raise ValueError("prompt_logprobs is not available with streaming")
        return self


class CompletionRequest(_SamplingMixin):
    model: Optional[str] = None
    prompt: Optional[Union[list[int], list[list[int]], str, list[str]]] = None
    echo: Optional[bool] = False
    logprobs: Optional[int] = None
    suffix: Optional[str] = None
    add_special_tokens: bool = True
    return_tokens_as_token_ids: Optional[bool] = None

    def to_sampling_params(self, default_max_tokens: int, defaults: Optional[dict] = None) -> SamplingParams:
        sp = self._common_sampling(default_max_tokens, self.logprobs, defaults)
        if self.echo and self.logprobs is not None and sp.prompt_logprobs is None:
            sp.prompt_logprobs = self.logprobs
        return sp


class CompletionLogProbs(OpenAIBaseModel):
    text_offset: list[int] = Field(default_factory=list)
    token_logprobs: list[Optional[float]] = Field(default_factory=list)
    tokens: list[str] = Field(default_factory=list)
    top_logprobs: list[Optional[dict[str, float]]] = Field(default_factory=list)


class CompletionResponseChoice(OpenAIBaseModel):
    index: int
    text: str
    logprobs: Optional[CompletionLogProbs] = None
    finish_reason: Optional[str] = None
    stop_reason: Optional[Union[int, str]] = None
    prompt_logprobs: Optional[list[Optional[dict[int, Any]]]] = None


class CompletionResponse(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("cmpl"))
    object: str = "text_completion"
    created: int = Field(default_factory=lambda: int(time.time()))
    model: str
    choices: list[CompletionResponseChoice]
    usage: UsageInfo
    kv_transfer_params: Optional[dict[str, Any]] = None


class CompletionResponseStreamChoice(OpenAIBaseModel):
    index: int
    text: str
    logprobs: Optional[CompletionLogProbs] = None
    finish_reason: Optional[str] = None
    stop_reason: Optional[Union[int, str]] = None


class CompletionStreamResponse(OpenAIBaseModel):
    id: str
    object: str = "text_completion"
    created: int = Field(default_factory=lambda: int(time.time()))
    model: str
    choices: list[CompletionResponseStreamChoice]
    usage: Optional[UsageInfo] = None


# ---- chat ---------------------------------------------------------------------------
class FunctionCall(OpenAIBaseModel):
    name: str
    arguments: str


class ToolCall(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("chatcmpl-tool"))
    type: Literal["function"] = "function"
    function: FunctionCall


class DeltaFunctionCall(OpenAIBaseModel):
    name: Optional[str] = None
    arguments: Optional[str] = None


class DeltaToolCall(OpenAIBaseModel):
    id: Optional[str] = None
    type: Optional[Literal["function"]] = None
    index: int
    function: Optional[DeltaFunctionCall] = None


class FunctionDefinition(OpenAIBaseModel):
    name: str
    description: Optional[str] = None
    parameters: Optional[dict[str, Any]] = None


class ChatCompletionToolsParam(OpenAIBaseModel):
    type: Literal["function"] = "function"
    function: FunctionDefinition


class ChatCompletionNamedToolChoiceParam(OpenAIBaseModel):
    function: dict
    type: Literal["function"] = "function"


class ChatCompletionRequest(_SamplingMixin):
    messages: list[dict[str, Any]]
    model: Optional[str] = None
    logprobs: Optional[bool] = False
    top_logprobs: Optional[int] = 0
    max_completion_tokens: Optional[int] = None
    tools: Optional[list[ChatCompletionToolsParam]] = None
    tool_choice: Optional[Union[Literal["none", "auto", "required"], ChatCompletionNamedToolChoiceParam]] = "none"
    parallel_tool_calls: Optional[bool] = None
    echo: bool = False
    add_generation_prompt: bool = True
    continue_final_message: bool = False
    add_special_tokens: bool = False
    chat_template: Optional[str] = None
    chat_template_kwargs: Optional[dict[str, Any]] = None
    return_tokens_as_token_ids: Optional[bool] = None

    def to_sampling_params(self, default_max_tokens: int, defaults: Optional[dict] = None) -> SamplingParams:
        if self.max_completion_tokens is not None and self.max_tokens is None:
            self.max_tokens = self.max_completion_tokens
        lp = (self.top_logprobs or 0) if self.logprobs else None
        sp = self._common_sampling(default_max_tokens, lp, defaults)
        if isinstance(self.tool_choice, ChatCompletionNamedToolChoiceParam) and self.tools and sp.guided_decoding is None:
            name = self.tool_choice.function.get("name")
            for t in self.tools:
                if t.function.name == name and t.function.parameters:
                    sp.guided_decoding = GuidedDecodingParams(json=t.function.parameters)
        elif self.tool_choice == "required" and self.tools and sp.guided_decoding is None:
            sp.guided_decoding = GuidedDecodingParams(json=self.required_tools_schema())
        return sp

    def required_tools_schema(self) -> dict:
        """``tool_choice="required"``: the reply must BE one or more calls of the given tools, so generation is
        constrained to a non-empty JSON array whose items are {"name": <a tool>, "parameters": <its schema>}
        (reference: vllm/entrypoints/openai/protocol.py:676-692)."""
        items, defs = [], {}
        for t in self.tools or []:
            params = dict(t.function.parameters or {"type": "object", "properties": {}})
            for key in ("$defs", "definitions"):
                for k, v in (params.pop(key, None) or {}).items():
                    if k in defs and defs[k] != v:
                        raise ValueError(f"tools define {k!r} in different ways")
                    defs[k] = v
            items.append({"type": "object",
                          "properties": {"name": {"type": "string", "enum": [t.function.name]}, "parameters": params},
                          "required": ["name", "parameters"]})
        schema: dict = {"type": "array", "minItems": 1, "items": {"anyOf": items} if len(items) != 1 else items[0]}
        if self.parallel_tool_calls is False:
            schema["maxItems"] = 1
        if defs:
            schema["$defs"] = defs
        return schema

    @model_validator(mode="after")
    def _check_chat(self):
        if self.top_logprobs and not self.logprobs:
            raise ValueError("top_logprobs needs logprobs=true")
        if self.tool_choice not in (None, "none") and not self.tools:
            raise ValueError("tool_choice needs tools")
        if self.tools and self.tool_choice is None:
            self.tool_choice = "auto"
        return self


class ChatMessage(OpenAIBaseModel):
    role: str
    content: Optional[str] = None
    reasoning_content: Optional[str] = None
    tool_calls: list[ToolCall] = Field(default_factory=list)


class ChatCompletionLogProb(OpenAIBaseModel):
    token: str
    logprob: float = -9999.0
    bytes: Optional[list[int]] = None


class ChatCompletionLogProbsContent(ChatCompletionLogProb):
    top_logprobs: list[ChatCompletionLogProb] = Field(default_factory=list)


class ChatCompletionLogProbs(OpenAIBaseModel):
    content: Optional[list[ChatCompletionLogProbsContent]] = None


class ChatCompletionResponseChoice(OpenAIBaseModel):
    index: int
    message: ChatMessage
    logprobs: Optional[ChatCompletionLogProbs] = None
    finish_reason: Optional[str] = "stop"
    stop_reason: Optional[Union[int, str]] = None


class ChatCompletionResponse(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("chatcmpl"))
    object: Literal["chat.completion"] = "chat.completion"
    created: int = Field(default_factory=lambda: int(time.time()))
    model: str
    choices: list[ChatCompletionResponseChoice]
    usage: UsageInfo
    prompt_logprobs: Optional[list[Optional[dict[int, Any]]]] = None
    kv_transfer_params: Optional[dict[str, Any]] = None


class DeltaMessage(OpenAIBaseModel):
    role: Optional[str] = None
    content: Optional[str] = None
    reasoning_content: Optional[str] = None
    tool_calls: list[DeltaToolCall] = Field(default_factory=list)


class ChatCompletionResponseStreamChoice(OpenAIBaseModel):
    index: int
    delta: DeltaMessage
    logprobs: Optional[ChatCompletionLogProbs] = None
    finish_reason: Optional[str] = None
    stop_reason: Optional[Union[int, str]] = None


class ChatCompletionStreamResponse(OpenAIBaseModel):
    id: str
    object: Literal["chat.completion.chunk"] = "chat.completion.chunk"
    created: int = Field(default_factory=lambda: int(time.time()))
    model: str
    choices: list[ChatCompletionResponseStreamChoice]
    usage: Optional[UsageInfo] = None


# ---- embeddings / pooling / score ------------------------------------------------------
class EmbeddingRequest(OpenAIBaseModel):
    model: Optional[str] = None
    input: Union[list[int], list[list[int]], str, list[str]]
    encoding_format: Literal["float", "base64"] = "float"
    dimensions: Optional[int] = None
    user: Optional[str] = None
    truncate_prompt_tokens: Optional[int] = None
    add_special_tokens: bool = True
    priority: int = 0

    def to_pooling_params(self) -> PoolingParams:
        return PoolingParams(dimensions=self.dimensions, task="embed")


class EmbeddingResponseData(OpenAIBaseModel):
    index: int
    object: str = "embedding"
    embedding: Union[list[float], str]


class EmbeddingResponse(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("embd"))
    object: str = "list"
    created: int = Field(default_factory=lambda: int(time.time()))
    model: str
    data: list[EmbeddingResponseData]
    usage: UsageInfo


class ScoreRequest(OpenAIBaseModel):
    model: Optional[str] = None
    text_1: Union[str, list[str]]
    text_2: Union[str, list[str]]
    truncate_prompt_tokens: Optional[int] = None


class ScoreResponseData(OpenAIBaseModel):
    index: int
    object: str = "score"
    score: float


class ScoreResponse(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("score"))
    object: str = "list"
    created: int = Field(default_factory=lambda: int(time.time()))
    model: str
    data: list[ScoreResponseData]
    usage: UsageInfo


class RerankRequest(OpenAIBaseModel):
    model: Optional[str] = None
    query: str
    documents: list[str]
    top_n: int = 0


class RerankDocument(OpenAIBaseModel):
    text: str


class RerankResult(OpenAIBaseModel):
    index: int
    document: RerankDocument
    relevance_score: float


class RerankUsage(OpenAIBaseModel):
    total_tokens: int


class RerankResponse(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("rerank"))
    model: str
    usage: RerankUsage
    results: list[RerankResult]


class PoolingRequest(EmbeddingRequest):
    """Raw pooler output of a pooling model (reference: /pooling)."""


class PoolingResponseData(OpenAIBaseModel):
    index: int
    object: str = "pooling"
    data: Any


class PoolingResponse(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("pool"))
    object: str = "list"
    created: int = Field(default_factory=lambda: int(time.time()))
    model: str
    data: list[PoolingResponseData]
    usage: UsageInfo


class ClassificationRequest(OpenAIBaseModel):
    model: Optional[str] = None
    input: Union[str, list[str]]
    truncate_prompt_tokens: Optional[int] = None
    priority: int = 0


class ClassificationData(OpenAIBaseModel):
    index: int
    label: Optional[str] = None
    probs: list[float]
    num_classes: int


class ClassificationResponse(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("classify"))
    object: str = "list"
    created: int = Field(default_factory=lambda: int(time.time()))
    model: str
    data: list[ClassificationData]
    usage: UsageInfo


# ---- Responses API ---------------------------------------------------------------------------
class ResponsesRequest(OpenAIBaseModel):
    model: Optional[str] = None
    input: Union[str, list[dict[str, Any]]]
    instructions: Optional[str] = None
    max_output_tokens: Optional[int] = None
    temperature: Optional[float] = None
    top_p: Optional[float] = None
    stream: bool = False
    store: bool = True
    background: bool = False
    metadata: Optional[dict[str, Any]] = None
    previous_response_id: Optional[str] = None
    user: Optional[str] = None
    priority: int = 0


class ResponseOutputText(OpenAIBaseModel):
    type: str = "output_text"
    text: str
    annotations: list = Field(default_factory=list)


class ResponseOutputMessage(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("msg"))
    type: str = "message"
    role: str = "assistant"
    status: str = "completed"
    content: list[ResponseOutputText]


class ResponseReasoningItem(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("rs"))
    type: str = "reasoning"
    summary: list = Field(default_factory=list)
    content: list[dict[str, Any]] = Field(default_factory=list)


class ResponseUsage(OpenAIBaseModel):
    input_tokens: int
    output_tokens: int
    total_tokens: int


class ResponsesResponse(OpenAIBaseModel):
    id: str = Field(default_factory=lambda: random_id("resp"))
    object: str = "response"
    created_at: int = Field(default_factory=lambda: int(time.time()))
    status: str = "completed"
    model: str
    output: list[Any]
    instructions: Optional[str] = None
    metadata: Optional[dict[str, Any]] = None
    temperature: Optional[float] = None
    top_p: Optional[float] = None
    max_output_tokens: Optional[int] = None
    previous_response_id: Optional[str] = None
    usage: Optional[ResponseUsage] = None
    incomplete_details: Optional[dict[str, Any]] = None


# ---- tokenizer ---------------------------------------------------------------------------
class TokenizeRequest(OpenAIBaseModel):
    model: Optional[str] = None
    prompt: Optional[str] = None
    messages: Optional[list[dict[str, Any]]] = None
    add_special_tokens: bool = True
    add_generation_prompt: bool = True
    return_token_strs: bool = False


class TokenizeResponse(OpenAIBaseModel):
    count: int
    max_model_len: int
    tokens: list[int]
    token_strs: Optional[list[str]] = None


class DetokenizeRequest(OpenAIBaseModel):
    model: Optional[str] = None
    tokens: list[int]


class DetokenizeResponse(OpenAIBaseModel):
    prompt: str


class LoadLoRAAdapterRequest(OpenAIBaseModel):
    lora_name: str
    lora_path: str


class UnloadLoRAAdapterRequest(OpenAIBaseModel):
    lora_name: str
    lora_int_id: Optional[int] = None


# ---- batch runner ----------------------------------------------------------------------------
class BatchRequestInput(OpenAIBaseModel):
    custom_id: str
    method: str
    url: str
    body: dict[str, Any]


class BatchResponseData(OpenAIBaseModel):
    status_code: int = 200
    request_id: str
    body: Optional[Any] = None


class BatchRequestOutput(OpenAIBaseModel):
    id: str
    custom_id: str
    response: Optional[BatchResponseData]
    error: Optional[Any]
