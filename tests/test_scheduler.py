"""Scheduler behaviour on CPU with hand-made model outputs
(modelled on the reference's tests/v1/core/test_scheduler.py)."""
import pytest

from tests.utils import (EOS, assert_scheduler_empty, make_requests, make_runner_output,
                         make_scheduler, step)
from vllm_distributed_amd.request import RequestStatus


def test_add_and_schedule_prefill():
    s = make_scheduler()
    reqs = make_requests(5)
    for r in reqs:
        s.add_request(r)
    assert len(s.waiting) == 5
    out = s.schedule()
    assert len(out.scheduled_new_reqs) == 5 and not out.scheduled_cached_reqs.req_ids
    assert out.total_num_scheduled_tokens == 50
    assert all(n == 10 for n in out.num_scheduled_tokens.values())
    assert len(s.running) == 5 and not s.waiting
    for d in out.scheduled_new_reqs:
        assert len(d.block_ids[0]) == 1 and d.num_computed_tokens == 0


def test_decode_after_prefill_and_finish_by_length():
    s = make_scheduler()
    (r,) = make_requests(1, num_tokens=20, max_tokens=3)
    s.add_request(r)
    n_out = 0
    for _ in range(10):
        if not s.has_unfinished_requests():
            break
        out, eco = step(s, token=5)
        for o in eco.get(0).outputs if eco else []:
            n_out += len(o.new_token_ids)
    assert n_out == 3
    assert r.status == RequestStatus.FINISHED_LENGTH_CAPPED
    s.schedule()  # flushes finished ids
    assert_scheduler_empty(s)


def test_chunked_prefill_budget():
    s = make_scheduler(max_num_batched_tokens=32, max_num_seqs=8)
    a, b = make_requests(2, num_tokens=40, max_tokens=4)
    s.add_request(a)
    s.add_request(b)
    o1 = s.schedule()
    assert o1.num_scheduled_tokens == {a.request_id: 32}
    s.update_from_output(o1, make_runner_output(o1, tokens_by_req={}))
    o2 = s.schedule()
    # a finishes its prompt (8 tokens), b gets the rest of the budget
    assert o2.num_scheduled_tokens == {a.request_id: 8, b.request_id: 24}
    s.update_from_output(o2, make_runner_output(o2, tokens_by_req={a.request_id: [1]}))
    o3 = s.schedule()
    assert o3.num_scheduled_tokens == {a.request_id: 1, b.request_id: 16}


def test_no_chunked_prefill_whole_prompts_only():
    s = make_scheduler(max_num_batched_tokens=2048, chunked=False, max_model_len=2048)
    a, b = make_requests(2, num_tokens=1500)
    s.add_request(a)
    s.add_request(b)
    o = s.schedule()
    assert list(o.num_scheduled_tokens) == [a.request_id]


def test_stop_via_eos_and_stop_token():
    s = make_scheduler()
    a, b = make_requests(2, max_tokens=10, stop_token_ids=[42])
    s.add_request(a)
    s.add_request(b)
    o = s.schedule()
    eco = s.update_from_output(o, make_runner_output(o, tokens_by_req={a.request_id: [EOS],
                                                                       b.request_id: [42]}))
    outs = {x.request_id: x for x in eco[0].outputs}
    assert str(outs[a.request_id].finish_reason) == "stop" and outs[a.request_id].stop_reason is None
    assert outs[b.request_id].stop_reason == 42
    assert not s.running
    s.schedule()
    assert_scheduler_empty(s)


def test_max_num_seqs_cap():
    s = make_scheduler(max_num_seqs=2)
    for r in make_requests(4):
        s.add_request(r)
    o = s.schedule()
    assert len(o.scheduled_new_reqs) == 2 and len(s.waiting) == 2


def test_preemption_recompute_and_resume():
    # 4 blocks usable (one is the null block): two 2-block requests fill the pool
    s = make_scheduler(num_blocks=5, block_size=16, max_model_len=256)
    a, b = make_requests(2, num_tokens=30, max_tokens=20, ignore_eos=True)
    s.add_request(a)
    s.add_request(b)
    step(s, token=1)   # prefill both, each now has 31 tokens -> 2 blocks each
    step(s, token=1)   # 32 tokens: still 2 blocks
    step(s, token=1)   # computes slot 31, the last one of block 2
    out = s.schedule()  # slot 32 needs a 3rd block: b (last) is preempted
    assert b.status == RequestStatus.PREEMPTED and b.num_computed_tokens == 0
    assert list(out.num_scheduled_tokens) == [a.request_id]
    assert s.waiting.peek_request() is b
    s.update_from_output(out, make_runner_output(out, token=1))
    # finish a -> b resumes from scratch and is flagged as resumed
    s.finish_requests(a.request_id, RequestStatus.FINISHED_ABORTED)
    out = s.schedule()
    assert out.scheduled_cached_reqs.req_ids == [b.request_id]
    assert out.scheduled_cached_reqs.resumed_from_preemption == [True]
    assert out.num_scheduled_tokens[b.request_id] == b.num_tokens
    assert a.request_id in out.finished_req_ids


def test_priority_policy_orders_and_preempts_lowest():
    s = make_scheduler(policy="priority", max_num_seqs=2)
    lo, hi, mid = make_requests(3, priorities=[5, 0, 3])
    for r in (lo, hi, mid):
        s.add_request(r)
    o = s.schedule()
    assert [d.req_id for d in o.scheduled_new_reqs] == [hi.request_id, mid.request_id]


def test_abort_running_and_waiting():
    s = make_scheduler(max_num_seqs=1)
    a, b = make_requests(2)
    s.add_request(a)
    s.add_request(b)
    s.schedule()
    s.finish_requests([a.request_id, b.request_id], RequestStatus.FINISHED_ABORTED)
    assert not s.running and not s.waiting
    o = s.schedule()
    assert o.finished_req_ids == {a.request_id, b.request_id}
    assert_scheduler_empty(s)


def test_prefix_cache_hit_second_request():
    s = make_scheduler(enable_prefix_caching=True)
    (a,) = make_requests(1, num_tokens=50, same_prompt=True, max_tokens=2)
    s.add_request(a)
    step(s)
    (b,) = make_requests(1, num_tokens=50, same_prompt=True, max_tokens=2)
    s.add_request(b)
    o = s.schedule()
    d = [x for x in o.scheduled_new_reqs if x.req_id == b.request_id][0]
    assert d.num_computed_tokens == 48          # three full blocks of 16 reused
    assert o.num_scheduled_tokens[b.request_id] == 2
    assert d.block_ids[0][:3] == s.kv_cache_manager.get_block_ids(a.request_id)[0][:3]
    stats = s.make_stats()
    assert stats.prefix_cache_stats.hits == 48


def test_spec_decode_accept_and_reject():
    s = make_scheduler(num_spec=3)
    (r,) = make_requests(1, num_tokens=8, max_tokens=32, ignore_eos=True)
    s.add_request(r)
    o = s.schedule()
    s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [1]},
                                               spec=[[11, 12, 13]]))
    assert r.spec_token_ids == [11, 12, 13]
    o = s.schedule()
    assert o.num_scheduled_tokens[r.request_id] == 4
    assert o.scheduled_spec_decode_tokens[r.request_id] == [11, 12, 13]
    # accept 1 draft (11), then the bonus/recovered token 99
    eco = s.update_from_output(o, make_runner_output(o, tokens_by_req={r.request_id: [11, 99]},
                                                     spec=[[21, 22, 23]]))
    assert r.output_token_ids == [1, 11, 99]
    assert r.num_computed_tokens == 8 + 2       # prompt + token "1" + accepted "11"
    st = eco[0].scheduler_stats.spec_decoding_stats
    assert st.num_draft_tokens == 3 and st.num_accepted_tokens == 1


def test_long_prefill_threshold():
    s = make_scheduler(long_prefill_token_threshold=16)
    (r,) = make_requests(1, num_tokens=100)
    s.add_request(r)
    assert s.schedule().num_scheduled_tokens[r.request_id] == 16


def test_pp_two_batches_in_flight():
    s = make_scheduler(pp=2)
    a, b = make_requests(2, num_tokens=10, max_tokens=4, ignore_eos=True)
    s.add_request(a)
    o1 = s.schedule()
    s.add_request(b)
    o2 = s.schedule()               # a's token is still in flight: only b is scheduled
    assert list(o2.num_scheduled_tokens) == [b.request_id]
    s.update_from_output(o1, make_runner_output(o1, token=3))
    o3 = s.schedule()
    assert list(o3.num_scheduled_tokens) == [a.request_id]
    assert o3.scheduled_cached_reqs.new_token_ids == [[3]]


def test_stats_and_events():
    s = make_scheduler()
    (r,) = make_requests(1, max_tokens=1)
    s.add_request(r)
    o = s.schedule()
    eco = s.update_from_output(o, make_runner_output(o))
    out = eco[0].outputs[0]
    assert [e.type.name for e in out.events] == ["QUEUED", "SCHEDULED"]
    assert eco[0].scheduler_stats.num_running_reqs == 0
