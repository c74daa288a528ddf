"""``vda`` / ``python -m vllm_distributed_amd`` command line:
serve | chat | complete | bench {latency,throughput,serve,prefix-caching,structured-output} | run-batch | collect-env.
Reference: vllm/entrypoints/cli/main.py, serve.py:35-249, openai.py, benchmark/*, run_batch.py."""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import sys


def _load_yaml_config(argv: list[str]) -> list[str]:
    """``--config file.yaml``: keys become flags placed BEFORE the explicit flags, so the
    command line wins."""
    if "--config" not in argv:
        return argv
    i = argv.index("--config")
    path = argv[i + 1]
    import yaml
    with open(path) as f:
        cfg = yaml.load(f, Loader=yaml.SafeLoader) or {}
    extra: list[str] = []
    for k, v in cfg.items():
        flag = "--" + str(k).replace("_", "-")
        if isinstance(v, bool):
            extra.append(flag if v else "--no-" + flag[2:])
        elif isinstance(v, (dict, list)):
            extra += [flag, json.dumps(v)]
        else:
            extra += [flag, str(v)]
    rest = argv[:i] + argv[i + 2:]
    # keep the subcommand (and positional model tag) first
    head = [a for a in rest[:2] if not a.startswith("-")]
    return head + extra + rest[len(head):]


def cmd_serve(args: argparse.Namespace) -> None:
    from ..openai.api_server import run_server
    if args.headless:
        from ...engine.headless import run_headless
        run_headless(args)
        return
    if args.api_server_count > 1:
        from ..openai.multi_api_server import run_multi_api_server
        run_multi_api_server(args)
        return
    asyncio.run(run_server(args))


def _client(args):
    import httpx
    headers = {}
    key = args.api_key or os.environ.get("OPENAI_API_KEY")
    if key:
        headers["Authorization"] = f"Bearer {key}"
    c = httpx.Client(base_url=args.url.rstrip("/"), headers=headers, timeout=None)
    model = args.model_name or c.get("/models").json()["data"][0]["id"]
    return c, model


def cmd_chat(args: argparse.Namespace) -> None:
    c, model = _client(args)
    conv = []
    if args.system_prompt:
        conv.append({"role": "system", "content": args.system_prompt})
    if args.quick:
        conv.append({"role": "user", "content": args.quick})
        r = c.post("/chat/completions", json={"model": model, "messages": conv}).json()
        print(r["choices"][0]["message"]["content"])
        return
    print("Please enter a message for the chat model:")
    while True:
        try:
            msg = input("> ")
        except EOFError:
            return
        conv.append({"role": "user", "content": msg})
        r = c.post("/chat/completions", json={"model": model, "messages": conv}).json()
        out = r["choices"][0]["message"]
        conv.append({"role": out["role"], "content": out["content"]})
        print(out["content"])


def cmd_complete(args: argparse.Namespace) -> None:
    c, model = _client(args)
    if args.quick:
        print(c.post("/completions", json={"model": model, "prompt": args.quick}).json()["choices"][0]["text"])
        return
    print("Please enter prompt to complete:")
    while True:
        try:
            p = input("> ")
        except EOFError:
            return
        print(c.post("/completions", json={"model": model, "prompt": p}).json()["choices"][0]["text"])


def cmd_collect_env(args: argparse.Namespace) -> None:
    from ...utils.collect_env import collect_env
    print(collect_env())


def cmd_run_batch(args: argparse.Namespace) -> None:
    from ..openai.run_batch import run_batch
    asyncio.run(run_batch(args))


def build_parser() -> argparse.ArgumentParser:
    from ...engine.arg_utils import AsyncEngineArgs
    from ..openai.api_server import make_arg_parser
    p = argparse.ArgumentParser(prog="vda", description="MI355X-native LLM serving engine")
    sub = p.add_subparsers(dest="cmd", required=True)

    sp = sub.add_parser("serve", help="start the OpenAI-compatible server")
    sp.add_argument("model_tag", type=str, nargs="?", default=None)
    make_arg_parser(sp)
    sp.set_defaults(fn=cmd_serve)

    for name, fn in (("chat", cmd_chat), ("complete", cmd_complete)):
        cp = sub.add_parser(name, help=f"{name} against a running server")
        cp.add_argument("--url", default="http://localhost:8000/v1")
        cp.add_argument("--model-name", default=None)
        cp.add_argument("--api-key", default=None)
        cp.add_argument("--system-prompt", default=None)
        cp.add_argument("-q", "--quick", default=None, help="send one message and exit")
        cp.set_defaults(fn=fn)

    bp = sub.add_parser("bench", help="benchmarks")
    bsub = bp.add_subparsers(dest="bench_cmd", required=True)
    from ...benchmarks import latency, prefix_caching, serve, structured_output, throughput
    for name, mod in (("latency", latency), ("throughput", throughput), ("serve", serve),
                      ("prefix-caching", prefix_caching), ("structured-output", structured_output)):
        x = bsub.add_parser(name)
        mod.add_cli_args(x)
        x.set_defaults(fn=mod.main)

    rb = sub.add_parser("run-batch", help="run a JSONL file of OpenAI batch requests offline")
    rb.add_argument("-i", "--input-file", required=True)
    rb.add_argument("-o", "--output-file", required=True)
    rb.add_argument("--model", type=str, required=True)
    AsyncEngineArgs.add_cli_args(rb)
    rb.set_defaults(fn=cmd_run_batch)

    ce = sub.add_parser("collect-env", help="print environment information")
    ce.set_defaults(fn=cmd_collect_env)
    return p


def main(argv=None) -> None:
    argv = _load_yaml_config(list(sys.argv[1:] if argv is None else argv))
    args = build_parser().parse_args(argv)
    args.fn(args)


if __name__ == "__main__":
    main()
