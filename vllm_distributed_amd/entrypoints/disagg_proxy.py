"""Front door of a disaggregated deployment: x prefill instances + y decode instances.

For every request the proxy (1) sends it to a prefill instance with ``max_tokens=1`` and
``kv_transfer_params={"do_remote_decode": true}``, (2) takes the ``kv_transfer_params`` of the
response (block ids + side-channel address of the prefill worker) and (3) forwards the original
request with those params to a decode instance, streaming its answer back unchanged.  Instances are
picked round-robin; the KV cache moves GPU to GPU (xGMI peer memory inside a node) without passing
through the proxy.
Reference: benchmarks/disagg_benchmarks/disagg_prefill_proxy_server.py,
tests/v1/kv_connector/nixl_integration/toy_proxy_server.py, examples/online_serving/disaggregated_serving.

    python -m vllm_distributed_amd.entrypoints.disagg_proxy --port 8000 \
        --prefill http://127.0.0.1:8100 --decode http://127.0.0.1:8200 http://127.0.0.1:8201
"""
from __future__ import annotations

import argparse
import itertools
import uuid
from typing import Optional

import httpx
from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse, StreamingResponse


def build_proxy(prefill_urls: list[str], decode_urls: list[str], transports: Optional[dict] = None) -> FastAPI:
    """``transports`` maps a base URL to an httpx transport (in-process apps in tests)."""
    app = FastAPI()
    transports = transports or {}
    clients = {u: httpx.AsyncClient(base_url=u, timeout=None, transport=transports.get(u))
               for u in prefill_urls + decode_urls}
    p_cycle, d_cycle = itertools.cycle(prefill_urls), itertools.cycle(decode_urls)

    async def handle(path: str, raw: Request):
        body = await raw.json()
        rid = raw.headers.get("X-Request-Id") or uuid.uuid4().hex
        headers = {"X-Request-Id": rid}
        if "authorization" in raw.headers:
            headers["Authorization"] = raw.headers["authorization"]
        pre = dict(body)
        pre.update(stream=False, max_tokens=1, kv_transfer_params={"do_remote_decode": True})
        pre.pop("stream_options", None)
        if "max_completion_tokens" in pre:
            pre["max_completion_tokens"] = 1
        r = await clients[next(p_cycle)].post(path, json=pre, headers=headers)
        if r.status_code != 200:
            return JSONResponse(r.json(), status_code=r.status_code)
        params = r.json().get("kv_transfer_params")
        dec = dict(body)
        if params:
            dec["kv_transfer_params"] = params        # without params the decode instance prefills itself
        target = clients[next(d_cycle)]
        if body.get("stream"):
            async def relay():
                async with target.stream("POST", path, json=dec, headers=headers) as resp:
                    async for chunk in resp.aiter_bytes():
                        yield chunk
            return StreamingResponse(relay(), media_type="text/event-stream")
        r2 = await target.post(path, json=dec, headers=headers)
        return JSONResponse(r2.json(), status_code=r2.status_code)

    @app.post("/v1/completions")
    async def completions(raw: Request):
        return await handle("/v1/completions", raw)

    @app.post("/v1/chat/completions")
    async def chat(raw: Request):
        return await handle("/v1/chat/completions", raw)

    @app.get("/v1/models")
    async def models():
        r = await clients[decode_urls[0]].get("/v1/models")
        return JSONResponse(r.json(), status_code=r.status_code)

    @app.get("/health")
    async def health():
        for u, c in clients.items():
            try:
                if (await c.get("/health")).status_code != 200:
                    return JSONResponse({"unhealthy": u}, status_code=503)
            except httpx.HTTPError:
                return JSONResponse({"unreachable": u}, status_code=503)
        return JSONResponse({})

    return app


def main(argv=None) -> None:
    import uvicorn
    ap = argparse.ArgumentParser(description="disaggregated prefill/decode proxy")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--prefill", nargs="+", required=True, help="base URLs of prefill instances")
    ap.add_argument("--decode", nargs="+", required=True, help="base URLs of decode instances")
    a = ap.parse_args(argv)
    uvicorn.run(build_proxy(a.prefill, a.decode), host=a.host, port=a.port)


if __name__ == "__main__":
    main()
