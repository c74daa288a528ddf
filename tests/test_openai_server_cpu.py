"""OpenAI-compatible server end to end on CPU: AsyncLLM (engine core in a child
process) + FastAPI app driven through httpx's ASGI transport (modelled on the
reference's tests/entrypoints/openai/*)."""
import asyncio
import json
import os

import httpx
import pytest

from tests.test_e2e_cpu import _save_tiny


def _add_tokenizer(path):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {f"w{i}": i for i in range(316)}
    vocab.update({"<unk>": 316, "<s>": 317, "</s>": 318, "<pad>": 319})
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", eos_token="</s>",
                                   pad_token="<pad>")
    fast.chat_template = ("{% for m in messages %}{{ m['role'] }} {{ m['content'] }} {% endfor %}"
                          "{% if add_generation_prompt %}assistant {% endif %}")
    fast.save_pretrained(path)


@pytest.fixture(scope="module")
def server(tmp_path_factory):
    from vllm_distributed_amd.entrypoints.openai.api_server import build_app, make_arg_parser
    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    tmp = str(tmp_path_factory.mktemp("srv"))
    path, _ = _save_tiny(tmp, "llama")
    _add_tokenizer(path)
    args = make_arg_parser().parse_args(["--model", path, "--dtype", "float32", "--device", "cpu",
                                         "--max-model-len", "256", "--max-num-seqs", "8",
                                         "--max-num-batched-tokens", "64", "--served-model-name", "tiny"])
    engine = AsyncLLM.from_engine_args(AsyncEngineArgs.from_cli_args(args))
    app = build_app(args, engine)
    # the engine's reader tasks live on one event loop: every test must run on it
    loop = asyncio.new_event_loop()
    yield app, loop
    engine.shutdown()
    loop.close()


def _run(server, coro_fn):
    app, loop = server

    async def go():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as c:
            return await asyncio.wait_for(coro_fn(c), timeout=120)
    return loop.run_until_complete(go())


def test_models_health_tokenize(server):
    async def f(c):
        assert (await c.get("/health")).status_code == 200
        m = (await c.get("/v1/models")).json()
        assert m["data"][0]["id"] == "tiny"
        t = (await c.post("/tokenize", json={"prompt": "w1 w2 w3", "add_special_tokens": False})).json()
        assert t["tokens"] == [1, 2, 3] and t["count"] == 3
        d = (await c.post("/detokenize", json={"tokens": [4, 5]})).json()
        assert d["prompt"] == "w4 w5"
        assert (await c.post("/v1/completions", json={"model": "nope", "prompt": "w1"})).status_code == 404
    _run(server, f)


def test_completion_and_streaming_agree(server):
    async def f(c):
        body = {"model": "tiny", "prompt": "w1 w2 w3 w4", "max_tokens": 8, "temperature": 0, "ignore_eos": True,
                "logprobs": 2}
        r = (await c.post("/v1/completions", json=body)).json()
        ch = r["choices"][0]
        assert r["usage"]["completion_tokens"] == 8 and ch["finish_reason"] == "length"
        assert len(ch["logprobs"]["tokens"]) == 8
        text = ""
        async with c.stream("POST", "/v1/completions", json={**body, "stream": True, "logprobs": None}) as s:
            async for line in s.aiter_lines():
                if line.startswith("data: ") and line != "data: [DONE]":
                    d = json.loads(line[6:])
                    if d["choices"]:
                        text += d["choices"][0]["text"]
        assert text == ch["text"]
        # two prompts, n=2
        r = (await c.post("/v1/completions", json={"prompt": ["w1 w2", "w9"], "max_tokens": 3, "n": 2,
                                                   "temperature": 0.8, "seed": 1})).json()
        assert len(r["choices"]) == 4
    _run(server, f)


def test_chat_completion(server):
    async def f(c):
        r = (await c.post("/v1/chat/completions", json={"model": "tiny", "max_tokens": 5, "temperature": 0,
                                                        "messages": [{"role": "user", "content": "w7 w8"}],
                                                        "logprobs": True, "top_logprobs": 2})).json()
        ch = r["choices"][0]
        assert ch["message"]["role"] == "assistant" and r["usage"]["completion_tokens"] == 5
        assert len(ch["logprobs"]["content"]) == 5 and len(ch["logprobs"]["content"][0]["top_logprobs"]) == 2
        n = 0
        async with c.stream("POST", "/v1/chat/completions", json={
                "messages": [{"role": "user", "content": "w7 w8"}], "max_tokens": 5, "temperature": 0,
                "stream": True, "stream_options": {"include_usage": True}}) as s:
            async for line in s.aiter_lines():
                if line.startswith("data: ") and line != "data: [DONE]":
                    d = json.loads(line[6:])
                    if d.get("usage"):
                        n = d["usage"]["completion_tokens"]
        assert n == 5
    _run(server, f)


def test_responses_api_and_invocations(server):
    async def f(c):
        chat = (await c.post("/v1/chat/completions", json={"model": "tiny", "max_tokens": 5, "temperature": 0,
                                                           "messages": [{"role": "user", "content": "w7 w8"}]})).json()
        r = (await c.post("/v1/responses", json={"model": "tiny", "input": "w7 w8", "max_output_tokens": 5,
                                                 "temperature": 0})).json()
        assert r["object"] == "response" and r["status"] == "incomplete"         # 5 tokens hit the limit
        text = r["output"][-1]["content"][0]["text"]
        assert text == chat["choices"][0]["message"]["content"] and r["usage"]["output_tokens"] == 5
        got = (await c.get(f"/v1/responses/{r['id']}")).json()
        assert got["id"] == r["id"]
        # a follow-up turn sees the first exchange
        r2 = (await c.post("/v1/responses", json={"input": [{"role": "user", "content": [{"type": "input_text",
                                                                                           "text": "w9"}]}],
                                                  "previous_response_id": r["id"], "max_output_tokens": 3,
                                                  "temperature": 0})).json()
        assert r2["previous_response_id"] == r["id"] and r2["usage"]["input_tokens"] > r["usage"]["input_tokens"]
        assert (await c.get("/v1/responses/resp_missing")).status_code == 404
        inv = (await c.post("/invocations", json={"model": "tiny", "prompt": "w1 w2", "max_tokens": 3,
                                                  "temperature": 0})).json()
        assert inv["object"] == "text_completion" and inv["usage"]["completion_tokens"] == 3
        assert (await c.post("/rerank", json={"query": "w1", "documents": ["w2"]})).status_code >= 400   # not a pooling model
    _run(server, f)


def test_metrics_endpoint(server):
    async def f(c):
        await c.post("/v1/completions", json={"prompt": "w1", "max_tokens": 2})
        txt = (await c.get("/metrics")).text
        assert "vllm:generation_tokens_total" in txt and "vllm:time_to_first_token_seconds" in txt
        await c.post("/v1/completions", json={"prompt": "w1", "max_tokens": 3, "n": 2, "temperature": 1.0})
        txt = (await c.get("/metrics")).text
        assert "vllm:cache_config_info{" in txt and 'block_size="16"' in txt
        assert "vllm:request_max_num_generation_tokens_count" in txt
        import re
        n_sum = re.search(r"vllm:request_params_n_sum\{[^}]*\} ([0-9.]+)", txt)
        assert n_sum and float(n_sum.group(1)) >= 3.0          # one request with n = 1 and one with n = 2
        assert (await c.get("/ping")).status_code == 200 and (await c.post("/ping")).status_code == 200
    _run(server, f)


def test_serving_benchmark_client(server):
    """The online benchmark client against the in-process app: all requests succeed and the
    summary has the streaming latency metrics."""
    import argparse
    from vllm_distributed_amd.benchmarks import serve as bs
    from vllm_distributed_amd.benchmarks.datasets import RandomDataset
    app, loop = server
    p = argparse.ArgumentParser()
    bs.add_cli_args(p)
    args = p.parse_args(["--model", "tiny", "--base-url", "http://t", "--num-prompts", "6", "--ignore-eos",
                         "--request-rate", "200", "--max-concurrency", "4", "--goodput", "ttft:60000"])
    reqs = RandomDataset(None, 0).sample(None, 6, input_len=12, output_len=5, vocab_size=300)
    res = loop.run_until_complete(asyncio.wait_for(
        bs.benchmark(args, reqs, transport=httpx.ASGITransport(app=app)), timeout=120))
    assert res["completed"] == 6 and res["failed"] == 0
    assert res["total_output_tokens"] == 30 and res["median_ttft_ms"] > 0 and res["request_goodput"] > 0


def test_disaggregated_proxy_end_to_end(tmp_path):
    """1 prefill + 1 decode API server behind the proxy (all in process, CPU): the proxied completion equals
    a plain one and the decode instance really imported the prompt's KV."""
    import socket
    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    from vllm_distributed_amd.entrypoints.disagg_proxy import build_proxy
    from vllm_distributed_amd.entrypoints.openai.api_server import build_app, make_arg_parser
    path, _ = _save_tiny(str(tmp_path), "llama")
    _add_tokenizer(path)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--model", path, "--dtype", "float32", "--device", "cpu", "--max-model-len", "256", "--max-num-seqs", "8",
              "--max-num-batched-tokens", "64", "--served-model-name", "tiny"]
    loop = asyncio.new_event_loop()
    engines, apps = [], {}
    try:
        for name, role in (("http://prefill", "kv_producer"), ("http://decode", "kv_consumer")):
            kt = json.dumps({"kv_connector": "XgmiConnector", "kv_role": role, "kv_port": port})
            args = make_arg_parser().parse_args(common + ["--kv-transfer-config", kt])
            eng = AsyncLLM.from_engine_args(AsyncEngineArgs.from_cli_args(args))
            engines.append(eng)
            apps[name] = httpx.ASGITransport(app=build_app(args, eng))
        proxy = build_proxy(["http://prefill"], ["http://decode"], transports=apps)
        prompt = " ".join(f"w{(7 * i) % 300}" for i in range(45))
        body = {"model": "tiny", "prompt": prompt, "max_tokens": 6, "temperature": 0, "ignore_eos": True}

        async def go():
            async with httpx.AsyncClient(transport=httpx.ASGITransport(app=proxy), base_url="http://proxy") as c:
                via = (await c.post("/v1/completions", json=body)).json()
            async with httpx.AsyncClient(transport=apps["http://prefill"], base_url="http://prefill") as c:
                plain = (await c.post("/v1/completions", json=body)).json()
            return via, plain
        via, plain = loop.run_until_complete(asyncio.wait_for(go(), 180))
        assert via["choices"][0]["text"] == plain["choices"][0]["text"]
        assert via["usage"]["completion_tokens"] == 6
    finally:
        for e in engines:
            e.shutdown()
        loop.close()


def test_audio_transcription_endpoint(tmp_path):
    """POST /v1/audio/transcriptions with a multipart WAV upload against a tiny Whisper: the text is what the
    engine generates from the task prompt for that clip (json and text response formats, 2 windows for a
    clip longer than the encoder window)."""
    import asyncio
    import io

    import httpx
    import numpy as np
    import torch
    import transformers as tf
    from scipy.io import wavfile
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    from vllm_distributed_amd.entrypoints.openai.api_server import build_app, make_arg_parser
    torch.manual_seed(0)
    cfg = tf.WhisperConfig(vocab_size=320, num_mel_bins=80, d_model=128, encoder_layers=1, encoder_attention_heads=2,
                           decoder_layers=1, decoder_attention_heads=2, encoder_ffn_dim=128, decoder_ffn_dim=128,
                           max_source_positions=1500, max_target_positions=32, pad_token_id=319, bos_token_id=317,
                           eos_token_id=318, decoder_start_token_id=310, suppress_tokens=None,
                           begin_suppress_tokens=None)
    path = os.path.join(tmp_path, "whisper")
    tf.WhisperForConditionalGeneration(cfg).float().eval().save_pretrained(path, safe_serialization=True)
    vocab = {f"w{i}": i for i in range(310)}
    vocab.update({"<|startoftranscript|>": 310, "<|en|>": 311, "<|transcribe|>": 312, "<|translate|>": 313,
                  "<|notimestamps|>": 314, "<|de|>": 315, "<unk>": 316, "<s>": 317, "</s>": 318, "<pad>": 319})
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", eos_token="</s>",
                            pad_token="<pad>").save_pretrained(path)
    args = make_arg_parser().parse_args(["--model", path, "--dtype", "float32", "--device", "cpu", "--max-model-len", "32",
                                         "--max-num-seqs", "4", "--max-num-batched-tokens", "64",
                                         "--served-model-name", "tiny-whisper"])
    engine = AsyncLLM.from_engine_args(AsyncEngineArgs.from_cli_args(args))
    app = build_app(args, engine)

    def wav(seconds):
        buf = io.BytesIO()
        wavfile.write(buf, 16000, (np.sin(np.arange(int(16000 * seconds)) / 9.0) * 12000).astype(np.int16))
        return buf.getvalue()

    async def go():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as c:
            a = await c.post("/v1/audio/transcriptions", files={"file": ("a.wav", wav(1.0), "audio/wav")},
                             data={"model": "tiny-whisper", "language": "en"})
            b = await c.post("/v1/audio/transcriptions", files={"file": ("a.wav", wav(1.0), "audio/wav")},
                             data={"model": "tiny-whisper", "response_format": "text"})
            long = await c.post("/v1/audio/translations", files={"file": ("b.wav", wav(31.0), "audio/wav")},
                                data={"language": "de", "response_format": "verbose_json"})
            bad = await c.post("/v1/audio/transcriptions", files={"file": ("a.wav", wav(0.5), "audio/wav")},
                               data={"language": "xx"})
            return a, b, long, bad
    loop = asyncio.new_event_loop()
    try:
        a, b, long, bad = loop.run_until_complete(asyncio.wait_for(go(), timeout=200))
    finally:
        engine.shutdown()
        loop.close()
    assert a.status_code == 200 and isinstance(a.json()["text"], str) and a.json()["text"]
    assert b.status_code == 200 and b.text == a.json()["text"]            # same clip, greedy
    assert long.status_code == 200 and long.json()["task"] == "translate" and long.json()["duration"] == 31.0
    assert bad.status_code == 400 and "xx" in bad.json()["message"]


# ---------------------------------------------------------------------------------------------------------
# streamed reasoning and tool calls: a character-level tokenizer lets the grammar steer the random tiny model into
# real JSON, so the whole path (structured output -> engine -> detokenizer -> incremental parsers -> SSE) runs
# ---------------------------------------------------------------------------------------------------------
def _add_char_tokenizer(path):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    chars = [chr(c) for c in range(32, 127)]
    vocab = {ch: i + 10 for i, ch in enumerate(chars)}       # ids 1 / 2 are the model config's bos / eos
    vocab.update({"<unk>": 0, "<s>": 1, "</s>": 2, "<pad>": 3})
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split("", "isolated")
    tok.decoder = decoders.Fuse()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", eos_token="</s>",
                                   pad_token="<pad>")
    fast.chat_template = ("{% for m in messages %}{{ m['role'] }}:{{ m['content'] }};{% endfor %}"
                          "{% if add_generation_prompt %}assistant:{% endif %}")
    fast.save_pretrained(path)


@pytest.fixture(scope="module")
def char_servers(tmp_path_factory):
    """One engine, two front ends: tools through the hermes parser; reasoning through the deepseek_r1 parser."""
    from vllm_distributed_amd.entrypoints.openai.api_server import build_app, make_arg_parser
    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    tmp = str(tmp_path_factory.mktemp("srvchar"))
    path, _ = _save_tiny(tmp, "llama")
    _add_char_tokenizer(path)
    base = ["--model", path, "--dtype", "float32", "--device", "cpu", "--max-model-len", "256", "--max-num-seqs", "8",
            "--max-num-batched-tokens", "64", "--served-model-name", "tiny"]
    args_t = make_arg_parser().parse_args(base + ["--enable-auto-tool-choice", "--tool-call-parser", "hermes"])
    args_r = make_arg_parser().parse_args(base + ["--reasoning-parser", "deepseek_r1"])
    engine = AsyncLLM.from_engine_args(AsyncEngineArgs.from_cli_args(args_t))
    loop = asyncio.new_event_loop()
    yield {"tools": (build_app(args_t, engine), loop), "reason": (build_app(args_r, engine), loop)}
    engine.shutdown()
    loop.close()


async def _sse(c, body):
    chunks = []
    async with c.stream("POST", "/v1/chat/completions", json=dict(body, stream=True)) as s:
        async for line in s.aiter_lines():
            if line.startswith("data: ") and line != "data: [DONE]":
                chunks.append(json.loads(line[6:]))
    return chunks


TOOLS = [{"type": "function", "function": {"name": "get_weather", "description": "weather",
                                           "parameters": {"type": "object",
                                                          "properties": {"city": {"type": "string", "enum": ["Paris", "Rome"]},
                                                                         "unit": {"type": "string", "enum": ["C", "F"]}},
                                                          "required": ["city", "unit"]}}},
         {"type": "function", "function": {"name": "get_time", "description": "time",
                                           "parameters": {"type": "object",
                                                          "properties": {"zone": {"type": "string", "enum": ["UTC", "CET"]}},
                                                          "required": ["zone"]}}}]


def test_required_tool_choice_streams_a_valid_call(char_servers):
    """tool_choice="required": the reply is grammar-constrained to a call of one of the tools; streamed, the call's
    id / name arrive first and the argument fragments concatenate to exactly the non-streamed arguments."""
    async def f(c):
        body = {"model": "tiny", "messages": [{"role": "user", "content": "weather?"}], "tools": TOOLS,
                "tool_choice": "required", "parallel_tool_calls": False, "max_tokens": 120, "temperature": 0}
        whole = (await c.post("/v1/chat/completions", json=body)).json()
        ch = whole["choices"][0]
        assert ch["finish_reason"] == "tool_calls", whole
        call = ch["message"]["tool_calls"][0]
        assert call["function"]["name"] in ("get_weather", "get_time") and not ch["message"].get("content")
        args = json.loads(call["function"]["arguments"])
        if call["function"]["name"] == "get_weather":
            assert args["city"] in ("Paris", "Rome") and args["unit"] in ("C", "F")
        else:
            assert args["zone"] in ("UTC", "CET")
        chunks = await _sse(c, body)
        name, frags, first_tool_chunk, ids = None, [], None, set()
        for k, d in enumerate(chunks):
            for tc in d["choices"][0]["delta"].get("tool_calls", []):
                assert tc["index"] == 0
                first_tool_chunk = k if first_tool_chunk is None else first_tool_chunk
                if tc.get("id"):
                    ids.add(tc["id"])
                    assert tc["type"] == "function" and name is None
                    name = tc["function"]["name"]
                if tc["function"].get("arguments"):
                    assert name is not None, "arguments before the name"
                    frags.append(tc["function"]["arguments"])
            assert not d["choices"][0]["delta"].get("content")
        assert name == call["function"]["name"] and len(ids) == 1
        assert "".join(frags) == call["function"]["arguments"]
        assert len(frags) >= 3, "the arguments were not streamed in pieces"
        assert first_tool_chunk < len(chunks) - 3, "the call only appeared when generation was over"
        assert chunks[-1]["choices"][0]["finish_reason"] == "tool_calls"
    _run(char_servers["tools"], f)


def test_named_tool_choice_streams_arguments(char_servers):
    async def f(c):
        body = {"model": "tiny", "messages": [{"role": "user", "content": "time?"}], "tools": TOOLS,
                "tool_choice": {"type": "function", "function": {"name": "get_time"}}, "max_tokens": 60,
                "temperature": 0}
        whole = (await c.post("/v1/chat/completions", json=body)).json()
        call = whole["choices"][0]["message"]["tool_calls"][0]
        assert call["function"]["name"] == "get_time" and json.loads(call["function"]["arguments"])["zone"] in ("UTC", "CET")
        chunks = await _sse(c, body)
        frags = [tc["function"].get("arguments") or "" for d in chunks
                 for tc in d["choices"][0]["delta"].get("tool_calls", [])]
        names = [tc["function"].get("name") for d in chunks for tc in d["choices"][0]["delta"].get("tool_calls", [])
                 if tc["function"].get("name")]
        assert names == ["get_time"] and "".join(frags) == call["function"]["arguments"] and len(frags) >= 3
    _run(char_servers["tools"], f)


def test_auto_tool_choice_streams_plain_content(char_servers):
    """"auto" with a parser configured: a reply without call syntax streams as ordinary content, chunk by chunk."""
    async def f(c):
        body = {"model": "tiny", "messages": [{"role": "user", "content": "hi"}], "tools": TOOLS, "tool_choice": "auto",
                "max_tokens": 12, "temperature": 0}
        whole = (await c.post("/v1/chat/completions", json=body)).json()
        chunks = await _sse(c, body)
        text = "".join(d["choices"][0]["delta"].get("content") or "" for d in chunks)
        if not whole["choices"][0]["message"].get("tool_calls"):
            assert text == (whole["choices"][0]["message"]["content"] or "")
            assert sum(1 for d in chunks if d["choices"][0]["delta"].get("content")) >= 3
    _run(char_servers["tools"], f)


def test_reasoning_is_streamed_while_the_model_generates(char_servers):
    """With a reasoning parser the stream must not wait for the end: reasoning_content deltas arrive token by token
    (the random model never closes its thinking, so everything is reasoning) and add up to the non-streamed field."""
    async def f(c):
        body = {"model": "tiny", "messages": [{"role": "user", "content": "think"}], "max_tokens": 16, "temperature": 0}
        whole = (await c.post("/v1/chat/completions", json=body)).json()
        msg = whole["choices"][0]["message"]
        chunks = await _sse(c, body)
        pieces = [(k, d["choices"][0]["delta"]["reasoning_content"]) for k, d in enumerate(chunks)
                  if d["choices"][0]["delta"].get("reasoning_content")]
        assert len(pieces) >= 4, "reasoning was not streamed in pieces"
        assert pieces[0][0] < len(chunks) - 4, "the first reasoning delta came only at the end"
        assert all(not d["choices"][0]["delta"].get("content") for d in chunks[1:])
        assert "".join(p for _, p in pieces).strip() == (msg.get("reasoning_content") or "").strip()
        assert not msg.get("content")
    _run(char_servers["reason"], f)
