"""Rank program for the multi-process tests: runs LLM.generate with the external-launcher executor and writes the
generated token ids (``<out>.<rank>``) and their log-probabilities (``<out>.<rank>.lp``) as JSON.  With a fourth
argument (JSON list of continuations, one per prompt) the continuations are teacher-forced instead: the engine
scores prompt + continuation and the log-probability of every forced token is written."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VLLM_ENABLE_V1_MULTIPROCESSING"] = "0"
os.environ.setdefault("VLLM_LOGGING_LEVEL", "WARNING")

from vllm_distributed_amd import LLM, SamplingParams  # noqa: E402


def main():
    model, out_path = sys.argv[1], sys.argv[2]
    kw = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
    prompts = [[(i * 7 + j) % 300 for j in range(20 + 5 * i)] for i in range(5)]
    base = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
                max_num_batched_tokens=64, distributed_executor_backend="external_launcher")
    base.update(kw)          # GPU rehearsals override device / dtype / memory settings
    forced = json.loads(sys.argv[4]) if len(sys.argv) > 4 else None
    llm = LLM(model=model, **base)
    if forced is not None:
        outs = llm.generate(prompt_token_ids=[p + c for p, c in zip(prompts, forced)],
                            sampling_params=SamplingParams(temperature=0, max_tokens=1, ignore_eos=True,
                                                           prompt_logprobs=0))
        toks = forced
        lps = [[o.prompt_logprobs[len(p) + k][t].logprob for k, t in enumerate(c)]
               for o, p, c in zip(outs, prompts, forced)]
    else:
        outs = llm.generate(prompt_token_ids=prompts,
                            sampling_params=SamplingParams(temperature=0, max_tokens=10, ignore_eos=True, logprobs=0))
        toks = [list(o.outputs[0].token_ids) for o in outs]
        lps = [[o.outputs[0].logprobs[k][t].logprob for k, t in enumerate(ts)] for o, ts in zip(outs, toks)]
    with open(f"{out_path}.{os.environ['RANK']}", "w") as f:
        json.dump(toks, f)
    with open(f"{out_path}.{os.environ['RANK']}.lp", "w") as f:
        json.dump(lps, f)
    try:
        info = llm.llm_engine.engine_core.engine_core.model_executor.worker.comm_info()
        with open(f"{out_path}.{os.environ['RANK']}.comm", "w") as f:
            json.dump(info, f)
    except Exception:   # noqa: BLE001 - executors without an in-process worker
        pass
    llm.shutdown()          # closes communicators and process groups in order (Worker.shutdown)


if __name__ == "__main__":
    main()
