"""Disaggregated prefill/decode on the GPU: two processes, the decode one maps the prefill one's
KV arena through HIP IPC and pulls blocks with the transfer kernel (the xGMI data plane; on a
one-GPU box both processes share the device, the code path is the same)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_prefill_decode_over_peer_memory(tmp_path):
    from tests.test_e2e_cpu import _save_tiny
    path, _ = _save_tiny(str(tmp_path), "llama")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "scripts", "disagg_worker.py"), role, path,
                               str(port), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for role in ("producer", "consumer")]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace"))
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    c = json.load(open(tmp_path / "consumer.json"))
    assert c["peer_memory"], "the consumer fell back to the socket data plane"
    assert c["got"] == c["ref"]
    assert all(n >= 96 for n in c["cached"])
    assert json.load(open(tmp_path / "producer.json"))["freed"] > 0


def test_push_into_the_landing_pool_over_peer_memory(tmp_path):
    """PUT_ASYNC on the GPU: the producer maps the consumer's landing arena, copies blocks into it layer by layer
    on its push stream (24 device slots, the other 14 blocks go to the pinned host pool), and has its blocks
    back before the consumer decodes; the consumer moves slots into blocks and pulls nothing."""
    from tests.test_e2e_cpu import _save_tiny
    path, _ = _save_tiny(str(tmp_path), "llama")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "scripts", "disagg_worker.py"), role, path,
                               str(port), str(tmp_path), "push"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for role in ("producer", "consumer")]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace"))
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    c = json.load(open(tmp_path / "consumer.json"))
    pr = json.load(open(tmp_path / "producer.json"))
    assert pr["landing_mapped"], "the producer pushed through the socket instead of the mapped landing arena"
    assert c["got"] == c["ref"]
    assert all(n >= 96 for n in c["cached"])
    assert c["pulled"] == 0, "the consumer fell back to pulling"
    assert c["pool_free_after"][0] == 24
