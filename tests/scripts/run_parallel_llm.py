"""Rank program for the multi-process CPU tests: runs LLM.generate with the
external-launcher executor and prints the generated token ids as JSON."""
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
    llm = LLM(model=model, **base)
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=10, ignore_eos=True))
    toks = [list(o.outputs[0].token_ids) for o in outs]
    with open(f"{out_path}.{os.environ['RANK']}", "w") as f:
        json.dump(toks, f)
    llm.shutdown()          # closes communicators and process groups in order (Worker.shutdown)


if __name__ == "__main__":
    main()
