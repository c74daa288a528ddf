"""Peer-memory all-reduce: N processes sharing the one GPU of the test box map each
other's arenas through HIP IPC and must reproduce Gloo sums exactly on integer data,
eagerly and under hipGraph replay (modelled on the reference's
tests/distributed/test_custom_all_reduce.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_ranks(world, script, extra_env=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", VDA_COMM_SPIN_TIMEOUT_MS="5000")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "scripts", script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]
    print(outs[0])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_custom_allreduce_shared_gpu(world):
    _run_ranks(world, "car_worker.py")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_fused_allreduce_rmsnorm_shared_gpu(world):
    """all-reduce + residual add + RMSNorm in one kernel: the bits of the unfused pair, graph replay."""
    _run_ranks(world, "far_worker.py")


@pytest.mark.parametrize("world", [2, 4])
def test_quantised_allreduce_shared_gpu(world):
    """int8 / int4 two-shot all-reduce: error within the two-rounding-step bound, graph replay."""
    _run_ranks(world, "qr_worker.py")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_peer_all_to_all_shared_gpu(world):
    """all-to-all-v of rows through peer memory: bit-exact against the same exchange over gloo, the
    return trip, hipGraph replay with changing counts, expert-parallel dispatch / combine round trip."""
    _run_ranks(world, "a2a_worker.py")


def test_withheld_call_is_reported_not_served():
    """A rank that never makes its call: the peers' kernels time out, return, and the per-step check raises."""
    _run_ranks(2, "car_timeout_worker.py", {"VDA_COMM_SPIN_TIMEOUT_MS": "100"})
