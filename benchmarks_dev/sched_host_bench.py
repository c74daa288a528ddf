"""Host time of the scheduler per decode step (no model, no GPU): N running requests, each step schedules one token
for every request and applies one sampled token.  Prints microseconds per step for schedule() and
update_from_output(), with the decode fast path / native free list on and off (VDA_NATIVE_BLOCK_POOL,
--no-fast-path).  python benchmarks_dev/sched_host_bench.py [--reqs 32,256,1024] [--steps 200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tests.utils import make_requests, make_runner_output, make_scheduler      # noqa: E402


def run(n, steps, caching, fast):
    s = make_scheduler(num_blocks=n * 40 + 100, max_num_seqs=n, max_num_batched_tokens=max(8192, 64 * n),
                       enable_prefix_caching=caching, log_stats=False)
    if not fast:
        s.kv_cache_manager._fast_ok = False
    for r in make_requests(n, num_tokens=40, max_tokens=steps + 50, ignore_eos=True):
        s.add_request(r)
    # prefill
    for _ in range(3):
        out = s.schedule()
        s.update_from_output(out, make_runner_output(out, token=5))
    t_s = t_u = 0.0
    for _ in range(steps):
        a = time.perf_counter()
        out = s.schedule()
        b = time.perf_counter()
        mo = make_runner_output(out, token=5)
        c = time.perf_counter()
        s.update_from_output(out, mo)
        d = time.perf_counter()
        t_s += b - a
        t_u += d - c
        assert len(out.num_scheduled_tokens) == n
    return t_s / steps * 1e6, t_u / steps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reqs", default="32,256,1024")
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    print(f"{'requests':>8} {'prefix cache':>12} {'fast path':>9} {'native list':>11} {'schedule us':>12} {'update us':>10}")
    for n in (int(x) for x in a.reqs.split(",")):
        for caching in (False, True):
            for fast, native in ((False, False), (False, True), (True, True)):
                os.environ["VDA_NATIVE_BLOCK_POOL"] = "1" if native else "0"
                ts, tu = run(n, a.steps, caching, fast)
                print(f"{n:>8} {str(caching):>12} {str(fast):>9} {str(native):>11} {ts:>12.0f} {tu:>10.0f}", flush=True)


if __name__ == "__main__":
    main()
