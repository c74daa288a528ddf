"""Host-side profile of the serving loop: where the CPU time of a decode step goes.
    python benchmarks_dev/host_profile.py [model] [batch] [out_len]
Prints cProfile's top functions by cumulative and by own time for one round of decode-heavy work."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VLLM_ENABLE_V1_MULTIPROCESSING", "0")
os.environ.setdefault("VLLM_LOGGING_LEVEL", "WARNING")

import torch  # noqa: E402

from vllm_distributed_amd import LLM, SamplingParams  # noqa: E402
from vllm_distributed_amd.inputs import TokensPrompt  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
out_len = int(sys.argv[3]) if len(sys.argv) > 3 else 128
llm = LLM(model=model, load_format="dummy", skip_tokenizer_init=True, dtype="bfloat16", max_model_len=1200,
          max_num_seqs=max(batch, 8), max_num_batched_tokens=16384, enable_prefix_caching=False, seed=0)
eng = llm.llm_engine
sp = SamplingParams(temperature=0.0, max_tokens=out_len, ignore_eos=True)


def round_(tag):
    ids = torch.randint(10, 30000, (batch, 1024))
    for i, row in enumerate(ids):
        eng.add_request(f"{tag}-{i}", TokensPrompt(prompt_token_ids=row.tolist()), sp)
    n = 0
    while eng.has_unfinished_requests():
        eng.step()
        n += 1
    return n


round_("w")
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
steps = round_("p")
pr.disable()
dt = time.perf_counter() - t0
print(f"{steps} steps in {dt:.3f} s -> {dt / steps * 1e3:.3f} ms/step (profiler on)")
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
llm.shutdown()
