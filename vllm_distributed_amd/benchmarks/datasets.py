"""Benchmark request sources.  Every dataset yields ``SampleRequest``s; the random one
needs no files (there is no network here), the file-backed ones read local JSON/JSONL/text.
Reference: vllm/benchmarks/datasets.py (RandomDataset, ShareGPTDataset, SonnetDataset,
BurstGPTDataset, CustomDataset)."""
from __future__ import annotations

import json
import random
from dataclasses import dataclass
from typing import Any, Optional

import numpy as np


@dataclass
class SampleRequest:
    prompt: Any                      # str or {"prompt_token_ids": [...]}
    prompt_len: int
    expected_output_len: int
    multi_modal_data: Optional[dict] = None
    lora_request: Any = None


class BenchmarkDataset:
    def __init__(self, dataset_path: Optional[str] = None, random_seed: int = 0):
        self.dataset_path = dataset_path
        self.random_seed = random_seed
        self.rng = random.Random(random_seed)

    def sample(self, tokenizer, num_requests: int, **kw) -> list[SampleRequest]:
        raise NotImplementedError

    @staticmethod
    def oversample(reqs: list[SampleRequest], n: int, rng: random.Random) -> list[SampleRequest]:
        if reqs and len(reqs) < n:
            reqs = reqs + rng.choices(reqs, k=n - len(reqs))
        return reqs[:n]


class RandomDataset(BenchmarkDataset):
    """Uniform random token ids; lengths jittered by ``range_ratio``; optional shared prefix."""

    def sample(self, tokenizer, num_requests: int, input_len: int = 1024, output_len: int = 128,
               range_ratio: float = 0.0, prefix_len: int = 0, vocab_size: Optional[int] = None,
               as_token_ids: bool = True, **kw) -> list[SampleRequest]:
        g = np.random.default_rng(self.random_seed)
        vocab = vocab_size or (tokenizer.vocab_size if tokenizer is not None else 32000)
        lo_i, hi_i = int(input_len * (1 - range_ratio)), int(input_len * (1 + range_ratio))
        lo_o, hi_o = int(output_len * (1 - range_ratio)), int(output_len * (1 + range_ratio))
        prefix = g.integers(0, vocab, size=prefix_len).tolist()
        out = []
        for i in range(num_requests):
            il = int(g.integers(max(lo_i, 1), hi_i + 1))
            ol = int(g.integers(max(lo_o, 1), hi_o + 1))
            ids = prefix + ((g.integers(0, vocab) + i + np.arange(il)) % vocab).tolist()
            if as_token_ids or tokenizer is None:
                prompt: Any = {"prompt_token_ids": ids}
            else:
                prompt = tokenizer.decode(ids)
            out.append(SampleRequest(prompt, len(ids), ol))
        return out


class ShareGPTDataset(BenchmarkDataset):
    """Local ShareGPT-format JSON: [{"conversations": [{"value": ...}, {"value": ...}]}]."""

    def sample(self, tokenizer, num_requests: int, output_len: Optional[int] = None, **kw) -> list[SampleRequest]:
        with open(self.dataset_path) as f:
            data = json.load(f)
        data = [d for d in data if len(d.get("conversations", [])) >= 2]
        self.rng.shuffle(data)
        out = []
        for d in data:
            if len(out) >= num_requests:
                break
            prompt, completion = d["conversations"][0]["value"], d["conversations"][1]["value"]
            pl = len(tokenizer(prompt).input_ids)
            ol = output_len or len(tokenizer(completion).input_ids)
            if pl < 4 or ol < 4 or pl > 1024 or pl + ol > 2048:
                continue
            out.append(SampleRequest(prompt, pl, ol))
        return self.oversample(out, num_requests, self.rng)


class SonnetDataset(BenchmarkDataset):
    """Lines of a local text file; a shared prefix of lines then random lines up to input_len."""

    def sample(self, tokenizer, num_requests: int, input_len: int = 550, output_len: int = 150,
               prefix_len: int = 200, **kw) -> list[SampleRequest]:
        with open(self.dataset_path) as f:
            lines = [l for l in f.readlines() if l.strip()]
        toks = [len(tokenizer(l).input_ids) for l in lines]
        avg = max(sum(toks) / len(toks), 1)
        n_in, n_pre = max(int(input_len / avg), 1), int(prefix_len / avg)
        prefix = lines[:n_pre]
        out = []
        for _ in range(num_requests):
            body = self.rng.choices(lines, k=max(n_in - n_pre, 1))
            prompt = "Pick as many lines as you can from these poem lines:\n" + "".join(prefix + body)
            out.append(SampleRequest(prompt, len(tokenizer(prompt).input_ids), output_len))
        return out


class CustomDataset(BenchmarkDataset):
    """JSONL with {"prompt": ..., "output_tokens": optional}."""

    def sample(self, tokenizer, num_requests: int, output_len: int = 128, **kw) -> list[SampleRequest]:
        out = []
        with open(self.dataset_path) as f:
            for line in f:
                if not line.strip():
                    continue
                d = json.loads(line)
                p = d["prompt"]
                out.append(SampleRequest(p, len(tokenizer(p).input_ids), int(d.get("output_tokens", output_len))))
        self.rng.shuffle(out)
        return self.oversample(out, num_requests, self.rng)


class BurstGPTDataset(BenchmarkDataset):
    """CSV trace with 'Request tokens' and 'Response tokens' columns; prompts are synthetic ids."""

    def sample(self, tokenizer, num_requests: int, vocab_size: Optional[int] = None, **kw) -> list[SampleRequest]:
        import csv
        rows = []
        with open(self.dataset_path) as f:
            for r in csv.DictReader(f):
                if r.get("Model", "GPT-4") == "GPT-4" and int(r["Response tokens"]) > 0:
                    rows.append((int(r["Request tokens"]), int(r["Response tokens"])))
        self.rng.shuffle(rows)
        vocab = vocab_size or tokenizer.vocab_size
        out = []
        for i, (il, ol) in enumerate(rows[:num_requests]):
            out.append(SampleRequest({"prompt_token_ids": [(i + j) % vocab for j in range(il)]}, il, ol))
        return self.oversample(out, num_requests, self.rng)


DATASETS = {"random": RandomDataset, "sharegpt": ShareGPTDataset, "sonnet": SonnetDataset,
            "custom": CustomDataset, "burstgpt": BurstGPTDataset}


def add_dataset_args(p, with_seed: bool = True) -> None:
    p.add_argument("--dataset-name", default="random", choices=sorted(DATASETS))
    p.add_argument("--dataset-path", default=None)
    p.add_argument("--num-prompts", type=int, default=1000)
    p.add_argument("--input-len", "--random-input-len", dest="input_len", type=int, default=1024)
    p.add_argument("--output-len", "--random-output-len", dest="output_len", type=int, default=128)
    p.add_argument("--random-range-ratio", type=float, default=0.0)
    p.add_argument("--prefix-len", "--random-prefix-len", dest="prefix_len", type=int, default=0)
    if with_seed:      # engine arguments bring their own --seed
        p.add_argument("--seed", type=int, default=0)


def get_samples(args, tokenizer, vocab_size: Optional[int] = None) -> list[SampleRequest]:
    ds = DATASETS[args.dataset_name](args.dataset_path, args.seed or 0)
    return ds.sample(tokenizer, args.num_prompts, input_len=args.input_len, output_len=args.output_len,
                     range_ratio=args.random_range_ratio, prefix_len=args.prefix_len, vocab_size=vocab_size)
