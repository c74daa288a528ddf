"""Scheduler / KV-cache scenarios on CPU with hand-made model outputs; every scenario ends with the leak
check ``assert_scheduler_empty``.  The scenario list follows the reference's tests/v1/core/test_scheduler.py,
test_prefix_caching.py, test_specialized_manager.py and tests/v1/kv_connector/unit/test_remote_*_lifecycle.py."""
import pytest
import torch

from tests.utils import (EOS, assert_scheduler_empty, make_config, make_requests, make_runner_output, make_scheduler,
                         step)
from vllm_distributed_amd.request import RequestStatus
from vllm_distributed_amd.sched.block_pool import BlockPool, FreeBlockQueue
from vllm_distributed_amd.sched.kv_cache_interface import (FullAttentionSpec, KVCacheConfig, KVCacheGroupSpec,
                                                           KVCacheTensor, SlidingWindowSpec)
from vllm_distributed_amd.sched.kv_cache_manager import KVCacheManager
from vllm_distributed_amd.sched.request_queue import FCFSQueue, PriorityQueue, create_request_queue
from vllm_distributed_amd.sched.scheduler import Scheduler


def drain(s, token=1, limit=500):
    """Run to completion; returns {request id: [tokens]} and the number of steps."""
    got, n = {}, 0
    while s.has_unfinished_requests():
        n += 1
        assert n < limit, "scheduler does not make progress"
        _, eco = step(s, token=token)
        for outs in (eco or {}).values():
            for o in outs.outputs:
                got.setdefault(o.request_id, []).extend(o.new_token_ids)
    s.schedule()            # flushes the finished ids to the (imaginary) workers
    return got, n


# ---------------------------------------------------------------------------------------------------------
# token budget, chunked prefill, ordering
# ---------------------------------------------------------------------------------------------------------
def test_running_requests_are_scheduled_before_waiting_ones():
    s = make_scheduler(max_num_batched_tokens=24)
    a, = make_requests(1, num_tokens=16, max_tokens=4)
    s.add_request(a)
    step(s)
    b, = make_requests(1, num_tokens=40, max_tokens=4)
    s.add_request(b)
    o = s.schedule()
    assert list(o.num_scheduled_tokens.items()) == [(a.request_id, 1), (b.request_id, 23)]
    s.update_from_output(o, make_runner_output(o, tokens_by_req={a.request_id: [1]}))
    drain(s)
    assert_scheduler_empty(s)


def test_budget_is_never_exceeded_under_mixed_load():
    s = make_scheduler(max_num_batched_tokens=50, max_num_seqs=8)
    for r in make_requests(6, num_tokens=33, max_tokens=5):
        s.add_request(r)
    while s.has_unfinished_requests():
        o = s.schedule()
        assert 0 < o.total_num_scheduled_tokens <= 50
        assert o.total_num_scheduled_tokens == sum(o.num_scheduled_tokens.values())
        toks = {rid: ([1] if s.requests[rid].num_computed_tokens >= s.requests[rid].num_tokens else [])
                for rid in o.num_scheduled_tokens}
        s.update_from_output(o, make_runner_output(o, tokens_by_req=toks))
    s.schedule()
    assert_scheduler_empty(s)


def test_every_request_gets_exactly_max_tokens():
    s = make_scheduler(max_num_batched_tokens=64, max_num_seqs=4)
    reqs = make_requests(7, num_tokens=21, max_tokens=6, ignore_eos=True)
    for r in reqs:
        s.add_request(r)
    got, _ = drain(s)
    assert {r.request_id: 6 for r in reqs} == {k: len(v) for k, v in got.items()}
    assert_scheduler_empty(s)


def test_prompt_longer_than_budget_takes_several_steps():
    s = make_scheduler(max_num_batched_tokens=16)
    r, = make_requests(1, num_tokens=50, max_tokens=2)
    s.add_request(r)
    sizes = []
    while r.num_computed_tokens < 50:
        o = s.schedule()
        sizes.append(o.num_scheduled_tokens[r.request_id])
        s.update_from_output(o, make_runner_output(o, tokens_by_req={
            r.request_id: [1] if r.num_computed_tokens >= r.num_tokens else []}))
    assert sizes == [16, 16, 16, 2]
    drain(s)
    assert_scheduler_empty(s)


def test_without_chunking_a_prompt_over_the_budget_waits_for_room():
    s = make_scheduler(max_num_batched_tokens=64, chunked=False, max_model_len=64)
    a, b = make_requests(2, num_tokens=40, max_tokens=2)
    s.add_request(a)
    s.add_request(b)
    o = s.schedule()
    assert list(o.num_scheduled_tokens) == [a.request_id]         # 40 + 40 > 64: b waits whole
    s.update_from_output(o, make_runner_output(o, token=1))
    o = s.schedule()
    assert o.num_scheduled_tokens == {a.request_id: 1, b.request_id: 40}
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


def test_long_prefill_threshold_caps_each_chunk():
    s = make_scheduler(max_num_batched_tokens=256, long_prefill_token_threshold=32)
    r, = make_requests(1, num_tokens=100, max_tokens=1)
    s.add_request(r)
    o = s.schedule()
    assert o.num_scheduled_tokens[r.request_id] == 32
    s.update_from_output(o, make_runner_output(o, tokens_by_req={}))
    drain(s)
    assert_scheduler_empty(s)


def test_max_model_len_stops_generation():
    s = make_scheduler(max_model_len=24)
    r, = make_requests(1, num_tokens=20, max_tokens=100, ignore_eos=True)
    s.add_request(r)
    got, _ = drain(s)
    assert len(got[r.request_id]) == 4 and r.status == RequestStatus.FINISHED_LENGTH_CAPPED
    assert_scheduler_empty(s)


def test_stop_token_in_the_middle_of_a_multi_token_step():
    s = make_scheduler(num_spec=3)
    r, = make_requests(1, max_tokens=10, stop_token_ids=[99])
    s.add_request(r)
    o = s.schedule()
    eco = s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [5, 99, 7]}))
    out = eco[0].outputs[0]
    assert out.new_token_ids == [5, 99] and out.stop_reason == 99     # what follows the stop token is dropped
    s.schedule()
    assert_scheduler_empty(s)


def test_eos_ignored_when_asked():
    s = make_scheduler()
    r, = make_requests(1, max_tokens=3, ignore_eos=True)
    s.add_request(r)
    got, _ = drain(s, token=EOS)
    assert got[r.request_id] == [EOS] * 3
    assert_scheduler_empty(s)


def test_new_requests_join_a_running_batch():
    s = make_scheduler(max_num_seqs=8)
    first = make_requests(2, max_tokens=8, ignore_eos=True)
    for r in first:
        s.add_request(r)
    step(s)
    step(s)
    late = make_requests(2, max_tokens=3, ignore_eos=True)
    for r in late:
        s.add_request(r)
    o = s.schedule()
    assert {d.req_id for d in o.scheduled_new_reqs} == {r.request_id for r in late}
    assert set(o.scheduled_cached_reqs.req_ids) == {r.request_id for r in first}
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


# ---------------------------------------------------------------------------------------------------------
# preemption and priority
# ---------------------------------------------------------------------------------------------------------
def test_preempted_request_finishes_with_the_same_number_of_tokens():
    s = make_scheduler(num_blocks=7, block_size=16, max_model_len=256)
    reqs = make_requests(3, num_tokens=30, max_tokens=12, ignore_eos=True)
    for r in reqs:
        s.add_request(r)
    got, _ = drain(s)
    assert all(len(got[r.request_id]) == 12 for r in reqs)
    assert_scheduler_empty(s)


def test_preemption_frees_the_victims_blocks_at_once():
    s = make_scheduler(num_blocks=5, block_size=16, max_model_len=256)
    a, b = make_requests(2, num_tokens=30, max_tokens=20, ignore_eos=True)
    s.add_request(a)
    s.add_request(b)
    for _ in range(3):
        step(s, token=1)
    s.schedule()
    assert b.status == RequestStatus.PREEMPTED
    assert not s.kv_cache_manager.get_block_ids(b.request_id)[0]
    assert len(s.kv_cache_manager.get_block_ids(a.request_id)[0]) == 3


def test_priority_preempts_an_already_scheduled_low_priority_victim():
    """Lower number = more urgent.  When the pool runs dry the least urgent running request goes, even
    though it was scheduled earlier in the same pass."""
    s = make_scheduler(num_blocks=5, block_size=16, max_model_len=256, policy="priority")
    lo, hi = make_requests(2, num_tokens=30, max_tokens=20, ignore_eos=True, priorities=[5, 0])
    s.add_request(lo)
    s.add_request(hi)
    for _ in range(3):
        step(s, token=1)
    o = s.schedule()
    assert lo.status == RequestStatus.PREEMPTED and hi.status == RequestStatus.RUNNING
    assert list(o.num_scheduled_tokens) == [hi.request_id]
    s.update_from_output(o, make_runner_output(o, token=1))
    got, _ = drain(s)
    assert_scheduler_empty(s)


def test_priority_queue_orders_by_priority_then_arrival():
    s = make_scheduler(policy="priority", max_num_seqs=1)
    reqs = make_requests(4, max_tokens=1, priorities=[3, 1, 2, 1])
    for r in reqs:
        s.add_request(r)
    order = []
    while s.has_unfinished_requests():
        o, _ = step(s)
        order += [d.req_id for d in o.scheduled_new_reqs]
    assert order == [reqs[1].request_id, reqs[3].request_id, reqs[2].request_id, reqs[0].request_id]
    s.schedule()
    assert_scheduler_empty(s)


def test_request_queues():
    a, b, c = make_requests(3, priorities=[2, 0, 1])
    q = FCFSQueue()
    for r in (a, b, c):
        q.add_request(r)
    assert q.peek_request() is a and len(q) == 3
    q.prepend_request(c)
    assert q.pop_request() is c
    q.remove_requests([a, c])
    assert list(q) == [b]
    p = PriorityQueue()
    for r in (a, b, c):
        p.add_request(r)
    assert [p.pop_request() for _ in range(3)] == [b, c, a]
    assert isinstance(create_request_queue("fcfs"), FCFSQueue) and isinstance(create_request_queue("priority"), PriorityQueue)


# ---------------------------------------------------------------------------------------------------------
# aborts
# ---------------------------------------------------------------------------------------------------------
def test_abort_mid_prefill_frees_everything():
    s = make_scheduler(max_num_batched_tokens=16)
    r, = make_requests(1, num_tokens=60)
    s.add_request(r)
    step(s)
    s.finish_requests(r.request_id, RequestStatus.FINISHED_ABORTED)
    assert not s.has_unfinished_requests()
    s.schedule()
    assert_scheduler_empty(s)


def test_abort_of_unknown_or_finished_request_is_a_no_op():
    s = make_scheduler()
    r, = make_requests(1, max_tokens=1)
    s.add_request(r)
    drain(s)
    s.finish_requests([r.request_id, "never-existed"], RequestStatus.FINISHED_ABORTED)
    assert_scheduler_empty(s)


def test_abort_preempted_request():
    s = make_scheduler(num_blocks=5, block_size=16, max_model_len=256)
    a, b = make_requests(2, num_tokens=30, max_tokens=20, ignore_eos=True)
    s.add_request(a)
    s.add_request(b)
    for _ in range(3):
        step(s, token=1)
    o = s.schedule()
    assert b.status == RequestStatus.PREEMPTED
    s.update_from_output(o, make_runner_output(o, token=1))
    s.finish_requests(b.request_id, RequestStatus.FINISHED_ABORTED)
    drain(s)
    assert_scheduler_empty(s)


def test_finished_ids_reach_the_workers_exactly_once():
    s = make_scheduler()
    r, = make_requests(1, max_tokens=1)
    s.add_request(r)
    step(s)
    o = s.schedule()
    assert o.finished_req_ids == {r.request_id}
    assert not s.schedule().finished_req_ids
    assert_scheduler_empty(s)


# ---------------------------------------------------------------------------------------------------------
# prefix caching
# ---------------------------------------------------------------------------------------------------------
def test_prefix_hit_counts_whole_blocks_only_and_leaves_one_token():
    s = make_scheduler(enable_prefix_caching=True, block_size=16)
    a, = make_requests(1, num_tokens=48, max_tokens=1, same_prompt=True)
    s.add_request(a)
    drain(s)
    b, = make_requests(1, num_tokens=48, max_tokens=1, same_prompt=True)
    s.add_request(b)
    o = s.schedule()
    # 48 tokens = 3 full blocks cached, but the last token must be computed to produce logits
    assert o.scheduled_new_reqs[0].num_computed_tokens == 32 and o.num_scheduled_tokens[b.request_id] == 16
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


def test_prefix_cache_shares_blocks_between_live_requests():
    s = make_scheduler(enable_prefix_caching=True, block_size=16)
    a, = make_requests(1, num_tokens=40, max_tokens=8, same_prompt=True, ignore_eos=True)
    s.add_request(a)
    step(s, token=1)
    b, = make_requests(1, num_tokens=40, max_tokens=8, same_prompt=True, ignore_eos=True)
    s.add_request(b)
    step(s, token=1)
    ia, ib = (s.kv_cache_manager.get_block_ids(r.request_id)[0] for r in (a, b))
    assert ia[:2] == ib[:2] and ia[2] != ib[2]
    pool = s.kv_cache_manager.block_pool
    assert all(pool.blocks[i].ref_cnt == 2 for i in ia[:2])
    drain(s)
    assert_scheduler_empty(s)


def test_different_prompts_do_not_hit():
    s = make_scheduler(enable_prefix_caching=True)
    a, b = make_requests(2, num_tokens=40, max_tokens=1)
    s.add_request(a)
    drain(s)
    s.add_request(b)
    o = s.schedule()
    assert o.scheduled_new_reqs[0].num_computed_tokens == 0
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


def test_cached_blocks_are_evicted_least_recently_used_first():
    s = make_scheduler(enable_prefix_caching=True, num_blocks=7, block_size=16, max_model_len=128)
    old, = make_requests(1, num_tokens=33, max_tokens=1)          # 3 blocks, 2 of them full
    s.add_request(old)
    drain(s)
    new, = make_requests(1, num_tokens=33, max_tokens=1)
    s.add_request(new)
    drain(s)
    # a third prompt needs 3 of the 6 usable blocks: the blocks of `old` (freed first) go before those of `new`
    third, = make_requests(1, num_tokens=33, max_tokens=1)
    s.add_request(third)
    drain(s)
    again, = make_requests(1, num_tokens=33, max_tokens=1)
    again.prompt_token_ids[:] = new.prompt_token_ids
    again._all_token_ids[:33] = new.prompt_token_ids
    s.add_request(again)
    o = s.schedule()
    assert o.scheduled_new_reqs[0].num_computed_tokens == 32, "the newer prompt must still be cached"
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


def test_reset_prefix_cache_only_when_nothing_runs():
    s = make_scheduler(enable_prefix_caching=True)
    a, = make_requests(1, num_tokens=40, max_tokens=4, same_prompt=True, ignore_eos=True)
    s.add_request(a)
    step(s)
    assert not s.reset_prefix_cache()
    drain(s)
    assert s.reset_prefix_cache()
    b, = make_requests(1, num_tokens=40, max_tokens=1, same_prompt=True)
    s.add_request(b)
    o = s.schedule()
    assert o.scheduled_new_reqs[0].num_computed_tokens == 0
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


def test_prefix_stats_count_queries_and_hits():
    s = make_scheduler(enable_prefix_caching=True)
    for _ in range(2):
        r, = make_requests(1, num_tokens=48, max_tokens=1, same_prompt=True)
        s.add_request(r)
        drain(s)
    st = s.kv_cache_manager.prefix_cache_stats
    total_q = st.queries if st else 0
    ps = s.make_stats().prefix_cache_stats
    assert ps is None or ps.queries >= 0
    assert total_q >= 0


def test_preempted_request_reuses_its_own_cached_prefix():
    s = make_scheduler(enable_prefix_caching=True, num_blocks=6, block_size=16, max_model_len=256)
    a, b = make_requests(2, num_tokens=32, max_tokens=24, ignore_eos=True)
    s.add_request(a)
    s.add_request(b)
    while b.status != RequestStatus.PREEMPTED:
        step(s, token=1)
    s.finish_requests(a.request_id, RequestStatus.FINISHED_ABORTED)
    o = s.schedule()
    # b's two full prompt blocks were cached when it was preempted: the resume recomputes only what follows
    assert o.num_scheduled_tokens[b.request_id] < b.num_tokens
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


# ---------------------------------------------------------------------------------------------------------
# block pool and free list
# ---------------------------------------------------------------------------------------------------------
def test_free_block_queue_order_and_removal():
    q = FreeBlockQueue(6, skip=[0])
    assert q.get_all_free_blocks() == [1, 2, 3, 4, 5]
    assert q.popleft() == 1
    q.remove(3)
    assert q.get_all_free_blocks() == [2, 4, 5] and 3 not in q and 4 in q
    q.append(1)
    q.append_n([3])
    assert q.popleft_n(3) == [2, 4, 5] and q.get_all_free_blocks() == [1, 3]


def test_block_pool_never_hands_out_the_null_block_and_counts_references():
    p = BlockPool(5, enable_caching=True)
    got = p.get_new_blocks(4)
    assert sorted(b.block_id for b in got) == [1, 2, 3, 4] and p.get_num_free_blocks() == 0
    with pytest.raises(Exception):
        p.get_new_blocks(1)
    p.touch([got[:2]])
    p.free_blocks(got)
    assert p.get_num_free_blocks() == 2            # the touched pair is still referenced once
    p.free_blocks(got[:2])
    assert p.get_num_free_blocks() == 4 and abs(p.get_usage()) < 1e-9


def test_freed_blocks_come_back_tail_first():
    """A request's blocks are freed in reverse so that its tail is evicted before its (more shareable) head."""
    s = make_scheduler(num_blocks=9, block_size=16, max_model_len=256, enable_prefix_caching=True)
    r, = make_requests(1, num_tokens=64, max_tokens=1)
    s.add_request(r)
    step(s)
    ids = list(s.kv_cache_manager.get_block_ids(r.request_id)[0])
    drain(s)
    free = s.kv_cache_manager.block_pool.free_queue.get_all_free_blocks()
    tail = [b for b in free if b in ids]
    assert tail == list(reversed(ids))


def test_usage_follows_allocation():
    s = make_scheduler(num_blocks=11, block_size=16)
    assert s.kv_cache_manager.usage == 0
    r, = make_requests(1, num_tokens=40, max_tokens=2)
    s.add_request(r)
    step(s)
    assert abs(s.kv_cache_manager.usage - 3 / 10) < 1e-9
    drain(s)
    assert s.kv_cache_manager.usage == 0


# ---------------------------------------------------------------------------------------------------------
# sliding-window and hybrid KV groups
# ---------------------------------------------------------------------------------------------------------
def _sched_with_spec(spec_groups, num_blocks=64, **kw):
    cfg = make_config(**kw)
    groups = [KVCacheGroupSpec([f"layer{i}"], sp) for i, sp in enumerate(spec_groups)]
    kv = KVCacheConfig(num_blocks=num_blocks, kv_cache_tensors=[KVCacheTensor(0, [f"layer{i}" for i in range(len(groups))])],
                       kv_cache_groups=groups)
    return Scheduler(cfg, kv, log_stats=True)


def test_sliding_window_frees_blocks_that_left_the_window():
    sw = SlidingWindowSpec(16, 1, 1, torch.float32, sliding_window=32)
    s = _sched_with_spec([sw], max_model_len=512)
    r, = make_requests(1, num_tokens=20, max_tokens=150, ignore_eos=True)
    s.add_request(r)
    peak = 0
    while s.has_unfinished_requests():
        step(s, token=1)
        if r.request_id in s.requests:
            ids = s.kv_cache_manager.get_block_ids(r.request_id)[0]
            live = [b for b in ids if b != 0]            # blocks out of the window are replaced by the null block
            peak = max(peak, len(live))
    assert peak <= 4, f"a 32-token window needs at most 3-4 live blocks, saw {peak}"
    s.schedule()
    assert_scheduler_empty(s)


def test_sliding_window_request_outlives_a_pool_smaller_than_its_length():
    sw = SlidingWindowSpec(16, 1, 1, torch.float32, sliding_window=32)
    s = _sched_with_spec([sw], num_blocks=8, max_model_len=512)
    r, = make_requests(1, num_tokens=10, max_tokens=300, ignore_eos=True)      # 310 tokens = 20 blocks > 7
    s.add_request(r)
    got, _ = drain(s)
    assert len(got[r.request_id]) == 300 and r.status == RequestStatus.FINISHED_LENGTH_CAPPED
    assert_scheduler_empty(s)


def test_hybrid_groups_allocate_in_lock_step():
    full = FullAttentionSpec(16, 1, 1, torch.float32)
    sw = SlidingWindowSpec(16, 1, 1, torch.float32, sliding_window=32)
    s = _sched_with_spec([full, sw], max_model_len=512)
    r, = make_requests(1, num_tokens=50, max_tokens=40, ignore_eos=True)
    s.add_request(r)
    step(s)
    f, w = s.kv_cache_manager.get_block_ids(r.request_id)
    assert len(f) == len(w) == 4 and not set(f) & set(b for b in w if b)
    drain(s)
    assert_scheduler_empty(s)


def test_hybrid_prefix_hit_is_the_length_all_groups_agree_on():
    full = FullAttentionSpec(16, 1, 1, torch.float32)
    sw = SlidingWindowSpec(16, 1, 1, torch.float32, sliding_window=32)
    s = _sched_with_spec([full, sw], max_model_len=512, enable_prefix_caching=True)
    a, = make_requests(1, num_tokens=80, max_tokens=1, same_prompt=True)
    s.add_request(a)
    drain(s)
    b, = make_requests(1, num_tokens=80, max_tokens=1, same_prompt=True)
    s.add_request(b)
    o = s.schedule()
    hit = o.scheduled_new_reqs[0].num_computed_tokens
    assert hit % 16 == 0 and 0 < hit <= 64
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


# ---------------------------------------------------------------------------------------------------------
# speculative decoding
# ---------------------------------------------------------------------------------------------------------
def test_spec_tokens_are_scheduled_and_rejected_ones_rolled_back():
    s = make_scheduler(num_spec=3)
    r, = make_requests(1, num_tokens=10, max_tokens=20, ignore_eos=True)
    s.add_request(r)
    o = s.schedule()
    s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [1]}, spec=[[2, 3, 4]]))
    o = s.schedule()
    assert o.num_scheduled_tokens[r.request_id] == 4 and o.scheduled_spec_decode_tokens[r.request_id] == [2, 3, 4]
    # the verifier accepts the first draft only (2) and adds its own next token (9)
    s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [2, 9]}, spec=[[]]))
    assert r.num_computed_tokens == 12 and r.num_output_tokens == 3
    drain(s)
    assert_scheduler_empty(s)


def test_spec_tokens_never_push_past_max_tokens():
    s = make_scheduler(num_spec=4)
    r, = make_requests(1, num_tokens=10, max_tokens=3, ignore_eos=True)
    s.add_request(r)
    o = s.schedule()
    s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [1]}, spec=[[2, 3, 4, 5]]))
    o = s.schedule()
    eco = s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [2, 3, 4, 5, 6]}, spec=[[]]))
    assert eco[0].outputs[0].new_token_ids == [2, 3] and r.num_output_tokens == 3
    s.schedule()
    assert_scheduler_empty(s)


def test_spec_stats_are_reported():
    s = make_scheduler(num_spec=2)
    r, = make_requests(1, max_tokens=8, ignore_eos=True)
    s.add_request(r)
    o = s.schedule()
    s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [1]}, spec=[[2, 3]]))
    o = s.schedule()
    eco = s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [2, 3, 4]}, spec=[[]]))
    st = next(iter(eco.values())).scheduler_stats.spec_decoding_stats
    assert st is not None and st.num_draft_tokens == 2 and st.num_accepted_tokens == 2
    drain(s)
    assert_scheduler_empty(s)


# ---------------------------------------------------------------------------------------------------------
# structured output
# ---------------------------------------------------------------------------------------------------------
def test_request_waiting_for_its_grammar_does_not_block_the_queue():
    from vllm_distributed_amd.structured_output import StructuredOutputRequest
    s = make_scheduler()
    a, b = make_requests(2, max_tokens=2)
    a.structured_output_request = StructuredOutputRequest(a.sampling_params)
    a.status = RequestStatus.WAITING_FOR_FSM
    s.add_request(a)
    s.add_request(b)
    o = s.schedule()
    assert list(o.num_scheduled_tokens) == [b.request_id] and a.status == RequestStatus.WAITING_FOR_FSM
    s.update_from_output(o, make_runner_output(o, token=1))
    s.finish_requests(a.request_id, RequestStatus.FINISHED_ABORTED)
    drain(s)
    assert_scheduler_empty(s)


# ---------------------------------------------------------------------------------------------------------
# pipeline parallelism: several batches in flight
# ---------------------------------------------------------------------------------------------------------
def test_request_in_flight_is_not_scheduled_twice():
    s = make_scheduler(pp=2, max_num_seqs=4)
    a, = make_requests(1, num_tokens=10, max_tokens=4, ignore_eos=True)
    s.add_request(a)
    o1 = s.schedule()
    o2 = s.schedule()                        # the first batch has not come back yet
    assert a.request_id in o1.num_scheduled_tokens and not o2.total_num_scheduled_tokens
    s.update_from_output(o1, make_runner_output(o1, token=1))
    drain(s)
    assert_scheduler_empty(s)


# ---------------------------------------------------------------------------------------------------------
# remote KV lifecycle (disaggregated prefill / decode), scheduler side with a scripted connector
# ---------------------------------------------------------------------------------------------------------
class ScriptedConnector:
    """Behaves like the pull connector's scheduler half: remote-prefill requests load asynchronously,
    remote-decode requests keep their blocks until the worker reports them sent."""
    def __init__(self, block_size=16):
        self.bs = block_size
        self.allocated, self.finished = [], []

    def get_num_new_matched_tokens(self, request, num_computed_tokens):
        p = request.kv_transfer_params
        if p and p.get("do_remote_prefill"):
            return len(request.prompt_token_ids) - num_computed_tokens, True
        return 0, False

    def update_state_after_alloc(self, request, blocks, num_external_tokens):
        self.allocated.append((request.request_id, num_external_tokens))
        if request.kv_transfer_params:
            request.kv_transfer_params["do_remote_prefill"] = False

    def build_connector_meta(self, so):
        return None

    def request_finished(self, request, block_ids):
        self.finished.append((request.request_id, list(block_ids)))
        p = request.kv_transfer_params
        if p and p.get("do_remote_decode") and request.status == RequestStatus.FINISHED_LENGTH_CAPPED:
            return True, {"do_remote_prefill": True, "remote_block_ids": list(block_ids)}
        return False, None


def _remote_prefill_request(n=40, max_tokens=4):
    r, = make_requests(1, num_tokens=n, max_tokens=max_tokens, ignore_eos=True,
                       kv_transfer_params={"do_remote_prefill": True, "remote_block_ids": [1, 2, 3]})
    return r


def test_remote_prefill_parks_the_request_until_the_kv_arrived():
    c = ScriptedConnector()
    s = make_scheduler(connector=c)
    r = _remote_prefill_request()
    s.add_request(r)
    o = s.schedule()
    assert not o.total_num_scheduled_tokens and r.status == RequestStatus.WAITING_FOR_REMOTE_KVS
    assert len(s.kv_cache_manager.get_block_ids(r.request_id)[0]) == 3          # blocks are there to be filled
    o = s.schedule()
    assert not o.total_num_scheduled_tokens
    from vllm_distributed_amd.sched.output import ModelRunnerOutput
    s.update_from_output(o, ModelRunnerOutput(finished_recving={r.request_id}))
    o = s.schedule()
    # everything but the last prompt token came over the wire
    assert o.num_scheduled_tokens == {r.request_id: 1} and o.scheduled_new_reqs[0].num_computed_tokens == 39
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


def test_other_requests_run_while_one_waits_for_remote_kv():
    c = ScriptedConnector()
    s = make_scheduler(connector=c)
    r = _remote_prefill_request()
    local, = make_requests(1, max_tokens=3, ignore_eos=True)
    s.add_request(r)
    s.add_request(local)
    o = s.schedule()
    assert list(o.num_scheduled_tokens) == [local.request_id]
    s.update_from_output(o, make_runner_output(o, token=1, finished_recving={r.request_id}))
    o = s.schedule()
    assert set(o.num_scheduled_tokens) == {local.request_id, r.request_id}
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


def test_abort_while_waiting_for_remote_kv_keeps_the_blocks_until_the_transfer_is_over():
    from vllm_distributed_amd.sched.output import ModelRunnerOutput
    c = ScriptedConnector()
    s = make_scheduler(connector=c, num_blocks=8)
    r = _remote_prefill_request()
    s.add_request(r)
    s.schedule()
    free_parked = s.kv_cache_manager.get_num_free_blocks()
    s.finish_requests(r.request_id, RequestStatus.FINISHED_ABORTED)
    assert not s.has_unfinished_requests()
    # the copy may still be writing: nobody else may get these blocks yet
    assert s.kv_cache_manager.get_num_free_blocks() == free_parked
    other, = make_requests(1, num_tokens=70, max_tokens=1)                         # needs 5 blocks, only 4 free
    s.add_request(other)
    o = s.schedule()
    assert not o.total_num_scheduled_tokens
    s.update_from_output(o, ModelRunnerOutput(finished_recving={r.request_id}))
    assert s.kv_cache_manager.get_num_free_blocks() == free_parked + 3
    drain(s)
    assert_scheduler_empty(s)


def test_failed_transfer_falls_back_to_local_prefill():
    from vllm_distributed_amd.sched.output import ModelRunnerOutput
    c = ScriptedConnector()
    s = make_scheduler(connector=c)
    r = _remote_prefill_request()
    s.add_request(r)
    s.schedule()
    o = s.schedule()
    s.update_from_output(o, ModelRunnerOutput(finished_recving={r.request_id}, failed_recving={r.request_id}))
    o = s.schedule()
    assert o.num_scheduled_tokens == {r.request_id: 40} and o.scheduled_new_reqs[0].num_computed_tokens == 0
    s.update_from_output(o, make_runner_output(o, token=1))
    drain(s)
    assert_scheduler_empty(s)


def test_remote_decode_keeps_blocks_until_sent_and_returns_params():
    from vllm_distributed_amd.sched.output import ModelRunnerOutput
    c = ScriptedConnector()
    s = make_scheduler(connector=c)
    r, = make_requests(1, num_tokens=40, max_tokens=1, kv_transfer_params={"do_remote_decode": True})
    s.add_request(r)
    o = s.schedule()
    eco = s.update_from_output(o, make_runner_output(o, token=1))
    out = eco[0].outputs[0]
    ids = out.kv_transfer_params["remote_block_ids"]
    assert out.finish_reason is not None and len(ids) == 3
    used = s.kv_cache_manager.block_pool.num_gpu_blocks - 1 - s.kv_cache_manager.get_num_free_blocks()
    assert used == 3 and not s.has_unfinished_requests()
    o = s.schedule()
    s.update_from_output(o, ModelRunnerOutput(finished_sending={r.request_id}))
    s.schedule()
    assert_scheduler_empty(s)


def test_aborted_remote_decode_request_is_freed_at_once():
    c = ScriptedConnector()
    s = make_scheduler(connector=c)
    r, = make_requests(1, num_tokens=40, max_tokens=5, kv_transfer_params={"do_remote_decode": True})
    s.add_request(r)
    step(s)
    s.finish_requests(r.request_id, RequestStatus.FINISHED_ABORTED)
    s.schedule()
    assert_scheduler_empty(s)


def test_remote_prefill_with_a_locally_cached_prefix_asks_for_the_rest_only():
    c = ScriptedConnector()
    s = make_scheduler(connector=c, enable_prefix_caching=True)
    warm, = make_requests(1, num_tokens=40, max_tokens=1, same_prompt=True)
    s.add_request(warm)
    drain(s)
    r, = make_requests(1, num_tokens=40, max_tokens=2, same_prompt=True,
                       kv_transfer_params={"do_remote_prefill": True, "remote_block_ids": [1, 2, 3]})
    s.add_request(r)
    s.schedule()
    rid, n_ext = c.allocated[-1]
    assert rid == r.request_id and n_ext == 40 - 32          # two full blocks were here already
    s.finish_requests(r.request_id, RequestStatus.FINISHED_ABORTED)
    from vllm_distributed_amd.sched.output import ModelRunnerOutput
    s.update_from_output(s.schedule(), ModelRunnerOutput(finished_recving={r.request_id}))
    s.schedule()
    assert_scheduler_empty(s)


# ---------------------------------------------------------------------------------------------------------
# run-ahead scheduling: a preempted request whose last token was still in flight
# ---------------------------------------------------------------------------------------------------------
def test_run_ahead_preempted_request_finishing_on_its_in_flight_token():
    """Run-ahead mode schedules step N + 1 before step N's tokens are known.  If that schedule preempts a request
    and step N's token then ends it (EOS), the request must leave the waiting queue as well: otherwise the next
    schedule() admits a finished request (reference: stopped_preempted_reqs in Scheduler.update_from_output)."""
    s = make_scheduler(num_blocks=9, max_num_batched_tokens=64, max_num_seqs=4, async_scheduling=True)
    assert s.async_mode
    a, b = make_requests(2, num_tokens=30, max_tokens=40)          # 16-token blocks: two blocks each at first
    s.add_request(a)
    s.add_request(b)
    o0 = s.schedule()                                               # both prefills
    s.update_from_output(o0, make_runner_output(o0, tokens_by_req={a.request_id: [1], b.request_id: [1]}))
    # decode steps run ahead: o_next is scheduled before o_prev's tokens arrive
    o_prev = s.schedule()
    victim = None
    for _ in range(40):
        o_next = s.schedule()
        if b.status == RequestStatus.PREEMPTED or a.status == RequestStatus.PREEMPTED:
            victim = b if b.status == RequestStatus.PREEMPTED else a
            break
        s.update_from_output(o_prev, make_runner_output(o_prev, token=1))
        o_prev = o_next
    assert victim is not None, "the block budget never forced a preemption"
    assert victim.request_id in o_prev.num_scheduled_tokens          # its token of the previous step is in flight
    other = a if victim is b else b
    # that in-flight token is EOS for the victim
    s.update_from_output(o_prev, make_runner_output(o_prev, tokens_by_req={victim.request_id: [EOS],
                                                                          other.request_id: [1]}))
    assert victim.is_finished()
    assert victim.request_id not in s.requests
    assert all(r is not victim for r in s.running)
    assert victim not in list(s.waiting)
    s.update_from_output(o_next, make_runner_output(o_next, token=1))
    s.finish_requests([other.request_id], RequestStatus.FINISHED_ABORTED)
    o = s.schedule()                                                 # must not raise on a finished request
    assert victim.request_id not in o.num_scheduled_tokens
    assert_scheduler_empty(s)
