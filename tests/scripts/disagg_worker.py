"""Rank program of the GPU disaggregation test: `producer` prefills and serves its KV arena,
`consumer` pulls the blocks over peer memory and decodes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VLLM_ENABLE_V1_MULTIPROCESSING"] = "0"
os.environ.setdefault("VLLM_LOGGING_LEVEL", "WARNING")

from vllm_distributed_amd import LLM, SamplingParams  # noqa: E402

PROMPTS = [[(7 * i + 3 * j) % 300 for j in range(100 + 30 * i)] for i in range(4)]


def main():
    import faulthandler
    faulthandler.dump_traceback_later(70, exit=True)      # a hang ends with stacks instead of silence
    role, model, port, workdir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    push = len(sys.argv) > 5 and sys.argv[5] == "push"
    kw = dict(skip_tokenizer_init=True, dtype="bfloat16", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=256, gpu_memory_utilization=0.05, enforce_eager=True)
    kt = dict(kv_connector="XgmiConnector", kv_port=port)
    ktp = {"do_remote_decode": True}
    if push:
        # the consumer listens on port + 100; 24 device landing slots for 38 blocks: the rest lands on the host
        kt = dict(kv_connector="XgmiPushConnector", kv_port=port + (100 if role == "consumer" else 0),
                  kv_connector_extra_config={"landing_slots": 24, "mem_pool_size_gb": 0.05})
        ktp.update(push_host="127.0.0.1", push_port=port + 100)
    if role == "producer":
        llm = LLM(model=model, kv_transfer_config=dict(kt, kv_role="kv_producer"), **kw)
        open(os.path.join(workdir, "producer.up"), "w").close()
        t0 = time.time()
        while push and not os.path.exists(os.path.join(workdir, "consumer.ready")) and time.time() - t0 < 60:
            time.sleep(0.1)
        sp = SamplingParams(temperature=0, max_tokens=1, ignore_eos=True, extra_args={"kv_transfer_params": ktp})
        outs = llm.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
        with open(os.path.join(workdir, "params.json.tmp"), "w") as f:
            json.dump([o.kv_transfer_params for o in outs], f)
        os.replace(os.path.join(workdir, "params.json.tmp"), os.path.join(workdir, "params.json"))
        t0 = time.time()
        while not os.path.exists(os.path.join(workdir, "consumer.done")) and time.time() - t0 < 45:
            time.sleep(0.1)
        pool = llm.llm_engine.engine_core.engine_core.scheduler.kv_cache_manager.block_pool
        before = pool.get_num_free_blocks()
        llm.generate(prompt_token_ids=[[1, 2, 3]], sampling_params=SamplingParams(max_tokens=2))
        conn = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.kv_connector
        with open(os.path.join(workdir, "producer.json"), "w") as f:
            json.dump({"freed": pool.get_num_free_blocks() - before, "free_now": pool.get_num_free_blocks(),
                       "landing_mapped": any(getattr(p, "landing", None) is not None for p in conn._peers.values())}, f)
        llm.shutdown()
    else:
        t0 = time.time()              # one engine at a time measures its memory share
        while not os.path.exists(os.path.join(workdir, "producer.up")) and time.time() - t0 < 60:
            time.sleep(0.1)
        ref_llm = LLM(model=model, **kw)
        sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
        ref = [list(o.outputs[0].token_ids) for o in ref_llm.generate(prompt_token_ids=PROMPTS, sampling_params=sp)]
        ref_llm.shutdown()
        llm = LLM(model=model, kv_transfer_config=dict(kt, kv_role="kv_consumer"), **kw)
        open(os.path.join(workdir, "consumer.ready"), "w").close()
        t0 = time.time()
        while not os.path.exists(os.path.join(workdir, "params.json")) and time.time() - t0 < 60:
            time.sleep(0.1)
        params = json.load(open(os.path.join(workdir, "params.json")))
        sps = [SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, extra_args={"kv_transfer_params": p})
               for p in params]
        outs = llm.generate(prompt_token_ids=PROMPTS, sampling_params=sps)
        conn = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.kv_connector
        mapped = any(p.region is not None for p in conn._peers.values())
        extra = {}
        if push:
            extra = {"pool_free_after": list(conn.pool.num_free()), "pulled": len(conn._peers)}
        with open(os.path.join(workdir, "consumer.json"), "w") as f:
            json.dump({"ref": ref, "got": [list(o.outputs[0].token_ids) for o in outs], "peer_memory": mapped,
                       "cached": [o.num_cached_tokens for o in outs], **extra}, f)
        open(os.path.join(workdir, "consumer.done"), "w").close()
        llm.shutdown()


if __name__ == "__main__":
    main()
