"""Structured-output benchmark: the same prompts generated freely and under a JSON-schema / regex / choice
constraint; reports throughput of both, the grammar compile time and the fraction of constrained outputs
that validate.  Reference: benchmarks/benchmark_serving_structured_output.py (offline form)."""
from __future__ import annotations

import argparse
import json
import re
import time

import numpy as np

_SCHEMA = {"type": "object", "properties": {"name": {"type": "string", "maxLength": 16}, "age": {"type": "integer"},
                                            "tags": {"type": "array", "items": {"enum": ["a", "b", "c"]}, "maxItems": 3}},
           "required": ["name", "age"], "additionalProperties": False}


def add_cli_args(p: argparse.ArgumentParser) -> None:
    from ..engine.arg_utils import EngineArgs
    p.add_argument("--model", type=str, required=True)
    p.add_argument("--num-prompts", type=int, default=32)
    p.add_argument("--input-len", type=int, default=64)
    p.add_argument("--output-len", type=int, default=64)
    p.add_argument("--structure-type", default="json", choices=["json", "regex", "choice"])
    p.add_argument("--output-json", default=None)
    EngineArgs.add_cli_args(p)


def main(args: argparse.Namespace) -> dict:
    from ..engine.arg_utils import EngineArgs
    from ..entrypoints.llm import LLM
    from ..sampling_params import GuidedDecodingParams, SamplingParams
    llm = LLM.from_engine_args(EngineArgs.from_cli_args(args))
    tok = llm.get_tokenizer()
    if tok is None:
        raise ValueError("the structured-output benchmark needs the model's tokenizer")
    vocab = llm.llm_engine.model_config.get_vocab_size()
    g = np.random.default_rng(0)
    prompts = [{"prompt_token_ids": g.integers(10, vocab - 10, args.input_len).tolist()} for _ in range(args.num_prompts)]
    gd = {"json": GuidedDecodingParams(json=_SCHEMA), "regex": GuidedDecodingParams(regex=r"[A-Z][a-z]{2,8}-[0-9]{3}"),
          "choice": GuidedDecodingParams(choice=["yes", "no", "maybe"])}[args.structure_type]
    free = SamplingParams(temperature=0.8, max_tokens=args.output_len, seed=0)
    cons = SamplingParams(temperature=0.8, max_tokens=args.output_len, seed=0, guided_decoding=gd)
    res = {"structure_type": args.structure_type, "num_prompts": args.num_prompts}
    for name, sp in (("free", free), ("constrained", cons)):
        t0 = time.perf_counter()
        outs = llm.generate(prompts, sp, use_tqdm=False)
        dt = time.perf_counter() - t0
        n = sum(len(o.outputs[0].token_ids) for o in outs)
        res[f"{name}_tok_per_s"] = round(n / dt, 1)
        if name == "constrained":
            texts = [o.outputs[0].text for o in outs if o.outputs[0].finish_reason == "stop"]

            def ok(t: str) -> bool:
                if args.structure_type == "json":
                    try:
                        d = json.loads(t)
                        return isinstance(d, dict) and "name" in d and "age" in d
                    except ValueError:
                        return False
                if args.structure_type == "regex":
                    return re.fullmatch(r"[A-Z][a-z]{2,8}-[0-9]{3}", t) is not None
                return t in ("yes", "no", "maybe")
            res["finished"] = len(texts)
            res["valid_fraction_of_finished"] = round(sum(ok(t) for t in texts) / max(len(texts), 1), 3)
    print(json.dumps(res))
    if args.output_json:
        with open(args.output_json, "w") as f:
            json.dump(res, f, indent=2)
    llm.shutdown()
    return res
