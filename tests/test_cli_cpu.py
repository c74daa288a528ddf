"""Command line: argument plumbing, YAML config merge, offline batch runner, latency bench."""
import json

from tests.test_e2e_cpu import _save_tiny
from tests.test_openai_server_cpu import _add_tokenizer

COMMON = ["--dtype", "float32", "--device", "cpu", "--max-model-len", "128", "--max-num-seqs", "4",
          "--max-num-batched-tokens", "64"]


def test_yaml_config_merge(tmp_path):
    from vllm_distributed_amd.entrypoints.cli.main import _load_yaml_config, build_parser
    cfg = tmp_path / "c.yaml"
    cfg.write_text("port: 9001\nmax_model_len: 77\nenable_prefix_caching: false\n")
    argv = _load_yaml_config(["serve", "some/model", "--config", str(cfg), "--port", "9002"])
    args = build_parser().parse_args(argv)
    assert args.model_tag == "some/model" and args.port == 9002 and args.max_model_len == 77
    assert args.enable_prefix_caching is False


def test_run_batch(tmp_path):
    from vllm_distributed_amd.entrypoints.cli.main import main
    path, _ = _save_tiny(str(tmp_path), "llama")
    _add_tokenizer(path)
    inp, out = tmp_path / "in.jsonl", tmp_path / "out.jsonl"
    inp.write_text("\n".join(json.dumps(x) for x in [
        {"custom_id": "a", "method": "POST", "url": "/v1/completions",
         "body": {"prompt": "w1 w2", "max_tokens": 3, "temperature": 0}},
        {"custom_id": "b", "method": "POST", "url": "/v1/chat/completions",
         "body": {"messages": [{"role": "user", "content": "w3"}], "max_tokens": 2}},
        {"custom_id": "c", "method": "POST", "url": "/v1/embeddings", "body": {"input": "w1"}}]) + "\n")
    main(["run-batch", "-i", str(inp), "-o", str(out), "--model", path] + COMMON)
    rows = [json.loads(l) for l in out.read_text().splitlines()]
    assert [r["custom_id"] for r in rows] == ["a", "b", "c"]
    assert rows[0]["response"]["body"]["usage"]["completion_tokens"] == 3
    assert rows[1]["error"] is None, rows[1]
    assert rows[1]["response"]["body"]["choices"][0]["message"]["role"] == "assistant"
    assert rows[2]["error"] is not None          # generative model: no embeddings route


def test_bench_latency_and_throughput(tmp_path):
    from vllm_distributed_amd.entrypoints.cli.main import main, build_parser
    path, _ = _save_tiny(str(tmp_path), "llama")
    oj = tmp_path / "lat.json"
    main(["bench", "latency", "--model", path, "--input-len", "8", "--output-len", "4", "--batch-size", "2",
          "--num-iters-warmup", "1", "--num-iters", "2", "--output-json", str(oj), "--skip-tokenizer-init"] + COMMON)
    assert json.loads(oj.read_text())["avg_latency"] > 0
    oj2 = tmp_path / "tp.json"
    main(["bench", "throughput", "--model", path, "--num-prompts", "5", "--input-len", "10", "--output-len", "4",
          "--output-json", str(oj2), "--skip-tokenizer-init"] + COMMON)
    r = json.loads(oj2.read_text())
    assert r["num_requests"] == 5 and r["output_tokens_per_second"] > 0


def test_bench_subcommands_parse_and_ramp_up_rates():
    """The benchmark subcommands build their parsers (help strings expand cleanly) and the ramp-up schedule
    hits its end points."""
    import argparse

    from vllm_distributed_amd.benchmarks import latency, prefix_caching, serve, structured_output, throughput
    from vllm_distributed_amd.benchmarks.serve import _rate_at
    for mod in (latency, prefix_caching, structured_output, throughput, serve):
        p = argparse.ArgumentParser()
        mod.add_cli_args(p)
        assert "--model" in p.format_help()
    assert _rate_at(0, 11, 5.0, "linear", 1.0, 11.0) == 1.0 and _rate_at(10, 11, 5.0, "linear", 1.0, 11.0) == 11.0
    assert abs(_rate_at(5, 11, 5.0, "exponential", 1.0, 16.0) - 4.0) < 1e-9
    assert _rate_at(3, 11, 5.0, None, 1.0, 16.0) == 5.0
