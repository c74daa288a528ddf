"""OpenAI-compatible HTTP server (FastAPI + uvicorn).

Routes: /health /ping /version /v1/models /v1/completions /v1/chat/completions
/v1/embeddings /v1/score /tokenize /detokenize /metrics /load /reset_prefix_cache
/sleep /wake_up /is_sleeping /start_profile /stop_profile /v1/load_lora_adapter
/v1/unload_lora_adapter /collective_rpc (dev mode).
Reference: vllm/entrypoints/openai/api_server.py:419-1113 (routes), :1116-1300
(middleware, state), vllm/entrypoints/launcher.py (serve_http).
"""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import signal
import socket
import uuid
from contextlib import asynccontextmanager
from http import HTTPStatus
from typing import Any, Optional

from fastapi import FastAPI, Request
from fastapi.exceptions import RequestValidationError
from fastapi.middleware.cors import CORSMiddleware
from fastapi.responses import JSONResponse, PlainTextResponse, Response, StreamingResponse

from ... import envs
from ...engine.arg_utils import AsyncEngineArgs
from ...engine.async_llm import AsyncLLM
from ...engine.core_client import EngineDeadError
from ...logger import init_logger
from ...version import __version__
from .protocol import (ChatCompletionRequest, ClassificationRequest, CompletionRequest, DetokenizeRequest,
                       EmbeddingRequest, ErrorResponse, LoadLoRAAdapterRequest, PoolingRequest, RerankRequest,
                       ResponsesRequest, ScoreRequest, TokenizeRequest, UnloadLoRAAdapterRequest)
from .serving import (OpenAIServingChat, OpenAIServingCompletion, OpenAIServingEmbedding, OpenAIServingModels,
                      OpenAIServingPooling, OpenAIServingResponses, OpenAIServingScores, OpenAIServingTokenization,
                      OpenAIServingTranscription)

logger = init_logger(__name__)


class RequestLogger:
    def __init__(self, max_log_len: Optional[int] = None):
        self.max_log_len = max_log_len

    def log_inputs(self, request_id: str, prompt: Optional[str], n_tokens: int, params) -> None:
        if prompt is not None and self.max_log_len is not None:
            prompt = prompt[: self.max_log_len]
        logger.info("received request %s: prompt=%r tokens=%d params=%s", request_id, prompt, n_tokens, params)


def make_arg_parser(parser: Optional[argparse.ArgumentParser] = None) -> argparse.ArgumentParser:
    parser = parser or argparse.ArgumentParser(description="OpenAI-compatible API server")
    parser.add_argument("--host", type=str, default=None)
    parser.add_argument("--port", type=int, default=8000)
    parser.add_argument("--uvicorn-log-level", type=str, default="info")
    parser.add_argument("--allow-credentials", action="store_true")
    parser.add_argument("--allowed-origins", type=json.loads, default=["*"])
    parser.add_argument("--allowed-methods", type=json.loads, default=["*"])
    parser.add_argument("--allowed-headers", type=json.loads, default=["*"])
    parser.add_argument("--api-key", type=str, default=None)
    parser.add_argument("--lora-modules", type=str, nargs="+", default=None, help="name=path pairs")
    parser.add_argument("--chat-template", type=str, default=None)
    parser.add_argument("--response-role", type=str, default="assistant")
    parser.add_argument("--ssl-keyfile", type=str, default=None)
    parser.add_argument("--ssl-certfile", type=str, default=None)
    parser.add_argument("--root-path", type=str, default=None)
    parser.add_argument("--middleware", type=str, action="append", default=[])
    parser.add_argument("--return-tokens-as-token-ids", action="store_true")
    parser.add_argument("--enable-auto-tool-choice", action="store_true")
    parser.add_argument("--tool-call-parser", type=str, default=None)
    parser.add_argument("--max-log-len", type=int, default=None)
    parser.add_argument("--enable-request-id-headers", action="store_true")
    parser.add_argument("--api-server-count", "-asc", type=int, default=1)
    parser.add_argument("--headless", action="store_true")
    parser.add_argument("--data-parallel-start-rank", "-dpr", type=int, default=0)
    parser.add_argument("--config", type=str, default=None, help="YAML file with the same keys as the flags")
    parser.add_argument("--model", type=str, default=None)
    AsyncEngineArgs.add_cli_args(parser)
    return parser


def load_chat_template(path_or_text: Optional[str]) -> Optional[str]:
    if path_or_text is None:
        return None
    if os.path.isfile(path_or_text):
        with open(path_or_text) as f:
            return f.read()
    return path_or_text.encode().decode("unicode_escape")


def _resp(result, stream_ok: bool = True):
    if isinstance(result, ErrorResponse):
        return JSONResponse(content=result.model_dump(), status_code=result.code)
    if hasattr(result, "model_dump"):
        return JSONResponse(content=result.model_dump())
    if stream_ok and hasattr(result, "__aiter__"):
        return StreamingResponse(content=result, media_type="text/event-stream")
    return JSONResponse(content=result)


def build_app(args: argparse.Namespace, engine: AsyncLLM) -> FastAPI:
    @asynccontextmanager
    async def lifespan(app: FastAPI):
        task = None
        if engine.log_stats:
            async def _log():
                while True:
                    await asyncio.sleep(10.0)
                    await engine.do_log_stats()
            task = asyncio.create_task(_log())
        try:
            yield
        finally:
            if task is not None:
                task.cancel()

    app = FastAPI(lifespan=lifespan, root_path=args.root_path or "")
    app.add_middleware(CORSMiddleware, allow_origins=args.allowed_origins, allow_credentials=args.allow_credentials,
                       allow_methods=args.allowed_methods, allow_headers=args.allowed_headers)
    mc = engine.model_config
    names = mc.served_model_name if isinstance(mc.served_model_name, list) else [mc.served_model_name]
    lora_modules = []
    for item in args.lora_modules or []:
        if "=" in item:
            n, p = item.split("=", 1)
            lora_modules.append({"name": n, "path": p})
        else:
            lora_modules.append(json.loads(item))
    rl = None if args.disable_log_requests else RequestLogger(args.max_log_len)
    template = load_chat_template(args.chat_template)
    default_sampling = engine.processor.generation_config if mc.generation_config == "auto" else {}
    ds = {k: default_sampling[k] for k in ("temperature", "top_p", "top_k", "min_p", "repetition_penalty")
          if default_sampling.get(k) is not None}
    models = OpenAIServingModels(engine, mc, names, lora_modules)
    common = dict(request_logger=rl, return_tokens_as_token_ids=args.return_tokens_as_token_ids)
    st = app.state
    st.engine, st.models = engine, models
    gen = mc.runner_type == "generate"
    st.completion = OpenAIServingCompletion(engine, mc, models, default_sampling=ds, **common) if gen else None
    st.chat = OpenAIServingChat(engine, mc, models, chat_template=template, tool_parser=args.tool_call_parser,
                                reasoning_parser=getattr(args, "reasoning_parser", "") or None,
                                enable_auto_tools=args.enable_auto_tool_choice, default_sampling=ds,
                                response_role=args.response_role, **common) if gen else None
    st.embedding = OpenAIServingEmbedding(engine, mc, models, **common) if not gen else None
    st.score = OpenAIServingScores(engine, mc, models, embedding=st.embedding, **common) if not gen else None
    st.pooling = OpenAIServingPooling(engine, mc, models, embedding=st.embedding, scores=st.score, **common) \
        if not gen else None
    st.responses = OpenAIServingResponses(engine, mc, models, chat=st.chat, **common) if gen else None
    st.tokenization = OpenAIServingTokenization(engine, mc, models, chat_template=template, **common)
    speech = any(x == "WhisperForConditionalGeneration" for x in (mc.architectures or []))
    st.transcription = OpenAIServingTranscription(engine, mc, models, **common) if speech else None
    st.server_load = 0

    # ---- middleware ---------------------------------------------------------------
    api_key = args.api_key or envs.VLLM_API_KEY
    if api_key:
        @app.middleware("http")
        async def auth(request: Request, call_next):
            if request.method == "OPTIONS" or not request.url.path.startswith("/v1"):
                return await call_next(request)
            if request.headers.get("Authorization") != f"Bearer {api_key}":
                return JSONResponse(content={"error": "Unauthorized"}, status_code=401)
            return await call_next(request)

    if args.enable_request_id_headers:
        @app.middleware("http")
        async def add_request_id(request: Request, call_next):
            rid = request.headers.get("X-Request-Id") or uuid.uuid4().hex
            resp = await call_next(request)
            resp.headers["X-Request-Id"] = rid
            return resp

    for m in args.middleware:
        import importlib
        mod, name = m.rsplit(".", 1)
        obj = getattr(importlib.import_module(mod), name)
        if isinstance(obj, type):
            app.add_middleware(obj)
        else:
            app.middleware("http")(obj)

    @app.exception_handler(RequestValidationError)
    async def validation_handler(_, exc):
        err = ErrorResponse(message=str(exc), type="BadRequestError", code=HTTPStatus.BAD_REQUEST)
        return JSONResponse(err.model_dump(), status_code=HTTPStatus.BAD_REQUEST)

    @app.exception_handler(EngineDeadError)
    async def dead_handler(_, exc):
        if not envs.VLLM_KEEP_ALIVE_ON_ENGINE_DEATH:
            logger.critical("engine died; shutting the server down")
            os.kill(os.getpid(), signal.SIGTERM)
        return Response(status_code=HTTPStatus.INTERNAL_SERVER_ERROR)

    # ---- routes -----------------------------------------------------------------------
    @app.get("/health")
    async def health():
        await engine.check_health()
        return Response(status_code=200)

    @app.api_route("/ping", methods=["GET", "POST"])
    async def ping():
        await engine.check_health()
        return Response(status_code=200)

    @app.get("/load")
    async def load():
        return JSONResponse({"server_load": st.server_load})

    @app.get("/version")
    async def version():
        return JSONResponse({"version": __version__})

    @app.get("/v1/models")
    async def list_models():
        return JSONResponse((await models.show_available_models()).model_dump())

    def _need(handler, what):
        if handler is None:
            return ErrorResponse(message=f"the served model does not support {what}", type="BadRequestError", code=400)
        return None

    async def _tracked(coro_fn, request_obj, raw):
        st.server_load += 1
        try:
            result = await coro_fn(request_obj, raw)
        except EngineDeadError:
            st.server_load -= 1
            raise
        if hasattr(result, "__aiter__"):
            async def wrapped():
                try:
                    async for x in result:
                        yield x
                finally:
                    st.server_load -= 1
            return StreamingResponse(content=wrapped(), media_type="text/event-stream")
        st.server_load -= 1
        return _resp(result)

    @app.post("/v1/completions")
    async def completions(request: CompletionRequest, raw: Request):
        e = _need(st.completion, "the Completions API")
        return _resp(e) if e else await _tracked(st.completion.create_completion, request, raw)

    @app.post("/v1/chat/completions")
    async def chat_completions(request: ChatCompletionRequest, raw: Request):
        e = _need(st.chat, "the Chat Completions API")
        return _resp(e) if e else await _tracked(st.chat.create_chat_completion, request, raw)

    @app.post("/v1/embeddings")
    async def embeddings(request: EmbeddingRequest, raw: Request):
        e = _need(st.embedding, "the Embeddings API")
        return _resp(e) if e else await _tracked(st.embedding.create_embedding, request, raw)

    @app.post("/score")
    @app.post("/v1/score")
    async def score(request: ScoreRequest, raw: Request):
        e = _need(st.score, "the Score API")
        return _resp(e) if e else await _tracked(st.score.create_score, request, raw)

    @app.post("/pooling")
    async def pooling(request: PoolingRequest, raw: Request):
        e = _need(st.pooling, "the Pooling API")
        return _resp(e) if e else await _tracked(st.pooling.create_pooling, request, raw)

    @app.post("/classify")
    async def classify(request: ClassificationRequest, raw: Request):
        e = _need(st.pooling, "the Classification API")
        return _resp(e) if e else await _tracked(st.pooling.create_classify, request, raw)

    @app.post("/rerank")
    @app.post("/v1/rerank")
    @app.post("/v2/rerank")
    async def rerank(request: RerankRequest, raw: Request):
        e = _need(st.pooling, "the Rerank API")
        return _resp(e) if e else await _tracked(st.pooling.create_rerank, request, raw)

    @app.post("/v1/responses")
    async def responses(request: ResponsesRequest, raw: Request):
        e = _need(st.responses, "the Responses API")
        return _resp(e) if e else await _tracked(st.responses.create_responses, request, raw)

    @app.get("/v1/responses/{response_id}")
    async def get_response(response_id: str):
        e = _need(st.responses, "the Responses API")
        return _resp(e) if e else _resp(await st.responses.retrieve_responses(response_id))

    @app.post("/v1/responses/{response_id}/cancel")
    async def cancel_response(response_id: str):
        e = _need(st.responses, "the Responses API")
        return _resp(e) if e else _resp(await st.responses.cancel_responses(response_id))

    @app.post("/v1/audio/transcriptions")
    async def transcriptions(raw: Request):
        e = _need(st.transcription, "the Transcriptions API")
        if e:
            return _resp(e)
        out = await st.transcription.create(raw, "transcribe")
        return PlainTextResponse(out) if isinstance(out, str) else _resp(out)

    @app.post("/v1/audio/translations")
    async def translations(raw: Request):
        e = _need(st.transcription, "the Translations API")
        if e:
            return _resp(e)
        out = await st.transcription.create(raw, "translate")
        return PlainTextResponse(out) if isinstance(out, str) else _resp(out)

    @app.post("/invocations")
    async def invocations(raw: Request):
        """SageMaker-style single entry point: the body's fields pick the API."""
        body = await raw.json()
        if "messages" in body:
            return await chat_completions(ChatCompletionRequest(**body), raw)
        if "prompt" in body:
            return await completions(CompletionRequest(**body), raw)
        if "query" in body and "documents" in body:
            return await rerank(RerankRequest(**body), raw)
        if "text_1" in body:
            return await score(ScoreRequest(**body), raw)
        if "input" in body:
            if st.embedding is not None:
                return await embeddings(EmbeddingRequest(**body), raw)
            return await responses(ResponsesRequest(**body), raw)
        return JSONResponse({"error": "cannot tell which API this body is for"}, status_code=400)

    @app.post("/tokenize")
    async def tokenize(request: TokenizeRequest, raw: Request):
        return _resp(await st.tokenization.create_tokenize(request, raw))

    @app.post("/detokenize")
    async def detokenize(request: DetokenizeRequest, raw: Request):
        return _resp(await st.tokenization.create_detokenize(request, raw))

    @app.post("/v1/load_lora_adapter")
    async def load_lora(request: LoadLoRAAdapterRequest):
        r = await models.load_lora_adapter(request.lora_name, request.lora_path)
        return _resp(r) if isinstance(r, ErrorResponse) else Response(status_code=200, content=r)

    @app.post("/v1/unload_lora_adapter")
    async def unload_lora(request: UnloadLoRAAdapterRequest):
        r = await models.unload_lora_adapter(request.lora_name)
        return _resp(r) if isinstance(r, ErrorResponse) else Response(status_code=200, content=r)

    @app.get("/metrics")
    async def metrics():
        import prometheus_client
        return Response(prometheus_client.generate_latest(), media_type=prometheus_client.CONTENT_TYPE_LATEST)

    @app.post("/start_profile")
    async def start_profile():
        await engine.start_profile()
        return Response(status_code=200)

    @app.post("/stop_profile")
    async def stop_profile():
        await engine.stop_profile()
        return Response(status_code=200)

    @app.post("/reset_prefix_cache")
    async def reset_prefix_cache():
        await engine.reset_prefix_cache()
        return Response(status_code=200)

    @app.post("/sleep")
    async def sleep(raw: Request):
        await engine.sleep(int(raw.query_params.get("level", "1")))
        return Response(status_code=200)

    @app.post("/wake_up")
    async def wake_up(raw: Request):
        tags = raw.query_params.getlist("tags") or None
        await engine.wake_up(tags)
        return Response(status_code=200)

    @app.get("/is_sleeping")
    async def is_sleeping():
        return JSONResponse({"is_sleeping": await engine.is_sleeping()})

    if envs.VLLM_SERVER_DEV_MODE:
        @app.get("/server_info")
        async def server_info():
            """The engine configuration as the server sees it (development mode only: it names paths and options).
            Reference: vllm/entrypoints/openai/api_server.py:961."""
            cfg = await engine.get_vllm_config() if hasattr(engine, "get_vllm_config") else getattr(engine, "vllm_config", None)
            return JSONResponse({"vllm_config": str(cfg), "version": __version__})

        @app.post("/collective_rpc")
        async def collective_rpc(raw: Request):
            body = await raw.json()
            res = await engine.collective_rpc(body["method"], body.get("timeout"), tuple(body.get("args", ())),
                                              body.get("kwargs"))
            return JSONResponse({"results": [r if isinstance(r, (int, float, str, bool, type(None), list, dict))
                                             else repr(r) for r in res]})

    return app


async def run_server(args: argparse.Namespace, sock: Optional[socket.socket] = None, client_index: int = 0,
                     attach: Optional[dict] = None, on_ready=None) -> None:
    import uvicorn
    if args.model is None and getattr(args, "model_tag", None):
        args.model = args.model_tag
    if args.model is None:
        raise SystemExit("a model is required (positional tag or --model)")
    engine_args = AsyncEngineArgs.from_cli_args(args)
    engine = AsyncLLM.from_engine_args(engine_args, client_index=client_index, attach=attach)
    app = build_app(args, engine)
    config = uvicorn.Config(app, host=args.host or "0.0.0.0", port=args.port, log_level=args.uvicorn_log_level,
                            timeout_keep_alive=envs.VLLM_HTTP_TIMEOUT_KEEP_ALIVE, ssl_keyfile=args.ssl_keyfile,
                            ssl_certfile=args.ssl_certfile)
    server = uvicorn.Server(config)
    loop = asyncio.get_running_loop()
    task = loop.create_task(server.serve(sockets=[sock] if sock else None))
    if on_ready is not None:
        async def _announce():
            while not server.started and not task.done():
                await asyncio.sleep(0.05)
            on_ready()
        loop.create_task(_announce())

    def stop(*_):
        server.should_exit = True

    for s in (signal.SIGINT, signal.SIGTERM):
        try:
            loop.add_signal_handler(s, stop)
        except NotImplementedError:
            pass
    try:
        await task
    finally:
        engine.shutdown()


if __name__ == "__main__":
    p = make_arg_parser()
    asyncio.run(run_server(p.parse_args()))
