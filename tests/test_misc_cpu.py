"""Small runtime pieces: the stateless process group, plugin loading and the local usage record."""
import json
import multiprocessing as mp
import os
import socket


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spg_rank(rank, world, port, q):
    from vllm_distributed_amd.parallel.utils import StatelessProcessGroup
    pg = StatelessProcessGroup.create("127.0.0.1", port, rank, world)
    got = pg.broadcast_obj({"id": 7, "blob": b"xyz"} if rank == 0 else None, src=0)
    gathered = pg.all_gather_obj(rank * 10)
    if rank == 0:
        pg.send_obj(["to", 1], dst=1)
        msg = None
    elif rank == 1:
        msg = pg.recv_obj(src=0)
    else:
        msg = None
    pg.barrier()
    q.put((rank, got, gathered, msg))


def test_stateless_process_group_three_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_spg_rank, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, got, gathered, msg in res:
        assert got == {"id": 7, "blob": b"xyz"} and gathered == [0, 10, 20]
        assert msg == (["to", 1] if rank == 1 else None)


def test_general_plugins_run_once_and_respect_the_allow_list(monkeypatch):
    import importlib.metadata as md

    import vllm_distributed_amd.plugins as P
    calls = []

    class EP:
        def __init__(self, name):
            self.name = name

        def load(self):
            return lambda: calls.append(self.name)
    monkeypatch.setattr(md, "entry_points", lambda group=None: [EP("a"), EP("b")] if group == P.DEFAULT_PLUGINS_GROUP else [])
    monkeypatch.setenv("VLLM_PLUGINS", "b")
    monkeypatch.setattr(P, "_loaded", False)
    P.load_general_plugins()
    P.load_general_plugins()
    assert calls == ["b"]


def test_usage_record_is_local_and_opt_in(tmp_path, monkeypatch):
    from vllm_distributed_amd.usage import UsageContext, UsageMessage
    monkeypatch.setenv("VLLM_CONFIG_ROOT", str(tmp_path))
    assert UsageMessage().report_usage("LlamaForCausalLM", UsageContext.LLM_CLASS) is None      # off by default
    assert not os.path.exists(tmp_path / "usage_stats.json")
    monkeypatch.setenv("VLLM_NO_USAGE_STATS", "0")
    monkeypatch.setenv("VLLM_DO_NOT_TRACK", "0")
    rec = UsageMessage().report_usage("LlamaForCausalLM", UsageContext.LLM_CLASS, {"dtype": "bfloat16"})
    line = json.loads(open(tmp_path / "usage_stats.json").read().strip())
    assert line["model_architecture"] == "LlamaForCausalLM" and line["dtype"] == "bfloat16" and rec["uuid"] == line["uuid"]
