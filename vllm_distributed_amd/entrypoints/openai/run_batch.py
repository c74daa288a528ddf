"""Offline OpenAI batch runner: one JSONL request per line in, one JSONL response per line out.
Reference: vllm/entrypoints/openai/run_batch.py."""
from __future__ import annotations

import argparse
import asyncio
import json
import uuid

from ...engine.arg_utils import AsyncEngineArgs
from ...engine.async_llm import AsyncLLM
from .protocol import ChatCompletionRequest, CompletionRequest, EmbeddingRequest, ErrorResponse, ScoreRequest
from .serving import (OpenAIServingChat, OpenAIServingCompletion, OpenAIServingEmbedding, OpenAIServingModels,
                      OpenAIServingScores)


def _read(path: str) -> str:
    if path.startswith(("http://", "https://")):
        import httpx
        return httpx.get(path).text
    with open(path) as f:
        return f.read()


def _write(path: str, text: str) -> None:
    if path.startswith(("http://", "https://")):
        import httpx
        httpx.put(path, content=text.encode())
        return
    with open(path, "w") as f:
        f.write(text)


async def run_batch(args: argparse.Namespace) -> None:
    engine = AsyncLLM.from_engine_args(AsyncEngineArgs.from_cli_args(args))
    try:
        mc = engine.model_config
        names = mc.served_model_name if isinstance(mc.served_model_name, list) else [mc.served_model_name]
        models = OpenAIServingModels(engine, mc, names, [])
        gen = mc.runner_type == "generate"
        chat = OpenAIServingChat(engine, mc, models) if gen else None
        comp = OpenAIServingCompletion(engine, mc, models) if gen else None
        emb = OpenAIServingEmbedding(engine, mc, models) if not gen else None
        score = OpenAIServingScores(engine, mc, models, embedding=emb) if not gen else None
        routes = {"/v1/chat/completions": (ChatCompletionRequest, chat and chat.create_chat_completion),
                  "/v1/completions": (CompletionRequest, comp and comp.create_completion),
                  "/v1/embeddings": (EmbeddingRequest, emb and emb.create_embedding),
                  "/v1/score": (ScoreRequest, score and score.create_score)}

        async def one(line: str) -> dict:
            req = json.loads(line)
            out = {"id": f"vda-{uuid.uuid4().hex}", "custom_id": req.get("custom_id"), "response": None, "error": None}
            cls, fn = routes.get(req.get("url"), (None, None))
            if fn is None:
                out["error"] = {"message": f"unsupported url {req.get('url')!r} for this model", "code": 400}
                return out
            try:
                body = cls(**req["body"])
                if getattr(body, "stream", False):
                    raise ValueError("streaming is not supported in batch mode")
                res = await fn(body)
            except Exception as e:   # noqa: BLE001 - reported per line
                out["error"] = {"message": str(e), "code": 400}
                return out
            if isinstance(res, ErrorResponse):
                out["error"] = res.model_dump()
            else:
                out["response"] = {"status_code": 200, "request_id": f"vda-batch-{uuid.uuid4().hex}",
                                   "body": res.model_dump()}
            return out

        lines = [l for l in _read(args.input_file).splitlines() if l.strip()]
        results = await asyncio.gather(*(one(l) for l in lines))
        _write(args.output_file, "".join(json.dumps(r) + "\n" for r in results))
    finally:
        engine.shutdown()
