"""Small runtime pieces: the stateless process group, plugin loading and the local usage record."""
import json
import pytest
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


def test_ray_executor_drives_workers_through_the_actor_protocol(tmp_path, monkeypatch):
    """Ray is not installed here: a stand-in module with the same call shapes (remote classes, object refs,
    placement groups) runs the actors in-process, so the executor's placement, rank wiring and RPC fan-out
    are exercised end to end with a real worker and model."""
    import sys
    import types

    class Ref:
        def __init__(self, v):
            self.v = v

    class Method:
        def __init__(self, fn):
            self.fn = fn

        def remote(self, *a, **kw):
            return Ref(self.fn(*a, **kw))

    class Handle:
        def __init__(self, obj):
            self._obj = obj

        def __getattr__(self, k):
            return Method(getattr(self._obj, k))

    class RemoteCls:
        def __init__(self, cls):
            self.cls = cls

        def options(self, **kw):
            return self

        def remote(self, *a, **kw):
            return Handle(self.cls(*a, **kw))

    ray = types.ModuleType("ray")
    ray.is_initialized = lambda: True
    ray.init = lambda **kw: None
    ray.remote = lambda **kw: (lambda cls: RemoteCls(cls))
    ray.get = lambda refs, timeout=None: [r.v for r in refs] if isinstance(refs, list) else refs.v
    ray.kill = lambda h: None
    ray.get_gpu_ids = lambda: [0]
    util = types.ModuleType("ray.util")
    util.get_node_ip_address = lambda: "127.0.0.1"
    util.remove_placement_group = lambda pg: None
    pgm = types.ModuleType("ray.util.placement_group")

    class PG:
        def ready(self):
            return Ref(True)
    pgm.placement_group = lambda bundles, strategy="PACK": PG()
    ssm = types.ModuleType("ray.util.scheduling_strategies")
    ssm.PlacementGroupSchedulingStrategy = lambda **kw: None
    ray.util, util.placement_group, util.scheduling_strategies = util, pgm, ssm
    for name, mod in (("ray", ray), ("ray.util", util), ("ray.util.placement_group", pgm),
                      ("ray.util.scheduling_strategies", ssm)):
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setenv("VLLM_ENABLE_V1_MULTIPROCESSING", "0")

    import torch

    from tests.test_e2e_cpu import _save_tiny
    from vllm_distributed_amd import LLM, SamplingParams
    path, hf = _save_tiny(str(tmp_path), "llama")
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4,
              distributed_executor_backend="ray")
    ex = llm.llm_engine.engine_core.engine_core.model_executor
    assert type(ex).__name__ == "RayDistributedExecutor" and len(ex.workers) == 1
    p = [5, 17, 99, 3, 250, 11]
    out = llm.generate(prompt_token_ids=[p], sampling_params=SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    with torch.no_grad():
        ref = hf.generate(torch.tensor([p]), max_new_tokens=6, do_sample=False, pad_token_id=0,
                          eos_token_id=None)[0, len(p):].tolist()
    assert list(out[0].outputs[0].token_ids) == ref
    llm.shutdown()


def test_layerwise_profiler_attributes_time_to_the_module_tree(tmp_path):
    """Hooks + timers around every module: the tree rows add up (a parent's time covers its children, self
    time is the remainder), classes are summarised, and the tables export."""
    import json

    import torch

    from tests.test_e2e_cpu import _save_tiny
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.profiler import layerwise_profile
    path, _ = _save_tiny(str(tmp_path), "llama")
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4)
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    with layerwise_profile(model, num_running_seqs=2) as prof:
        llm.generate(prompt_token_ids=[[1, 2, 3, 4], [5, 6, 7]], sampling_params=SamplingParams(max_tokens=3))
    llm.shutdown()
    res = prof.results
    names = [e.name for e in res.model_stats]
    assert "model.layers.0.self_attn" in names and "model.layers.1.mlp.down_proj" in names
    by = {e.name: e for e in res.model_stats}
    layer = by["model.layers.0"]
    assert layer.calls == 3 and layer.device_time_us > 0
    kids = sum(e.device_time_us for e in res.model_stats if e.name.startswith("model.layers.0.") and e.depth == layer.depth + 1)
    assert kids <= layer.device_time_us * 1.001 and abs(layer.device_time_us - kids - layer.self_device_time_us) < 1.0
    classes = {e.name for e in res.summary_stats}
    assert {"LlamaDecoderLayer", "RMSNorm", "Attention"} <= classes
    assert abs(sum(e.pct_device_time for e in res.summary_stats) - 100.0) < 1.0
    res.export_model_stats_table_csv(str(tmp_path / "m.csv"))
    res.export_summary_stats_table_csv(str(tmp_path / "s.csv"))
    d = res.convert_stats_to_dict()
    json.dumps(d)
    assert d["metadata"]["num_running_seqs"] == 2 and d["model_stats"] and d["summary_stats"]
    import io
    buf = io.StringIO()
    res.print_model_table(max_depth=3, file=buf)
    res.print_summary_table(file=buf)
    assert "self_attn" in buf.getvalue() and "module class" in buf.getvalue()


def test_streamed_safetensors_loader(tmp_path):
    """The native range reader returns every tensor of a multi-file checkpoint bit for bit (windows smaller than the
    checkpoint, tensors larger than a read piece, odd dtypes), and an engine loaded through it decodes like one
    loaded the ordinary way."""
    import torch
    from safetensors.torch import save_file

    from tests.test_e2e_cpu import _save_tiny
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.models.loader import streamed_safetensors_iterator
    g = torch.Generator().manual_seed(0)
    a = {"big": torch.randn(3000, 3001, generator=g), "half": torch.randn(17, 5, generator=g).half(),
         "bf": torch.randn(129, generator=g).bfloat16(), "i8": torch.randint(-100, 100, (7, 9), dtype=torch.int8),
         "flag": torch.tensor([True, False, True]), "scalar": torch.tensor(3.5)}
    b = {"f8": torch.randn(64, 32, generator=g).to(torch.float8_e4m3fn), "idx": torch.arange(1000, dtype=torch.int64),
         "empty": torch.zeros(0, 4)}
    save_file(a, str(tmp_path / "a.safetensors"))
    save_file(b, str(tmp_path / "b.safetensors"))
    got = {}
    for name, t in streamed_safetensors_iterator([str(tmp_path / "a.safetensors"), str(tmp_path / "b.safetensors")],
                                                 concurrency=5, window_bytes=1 << 20):
        got[name] = t.clone()
    want = {**a, **b}
    assert set(got) == set(want)
    for k, v in want.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        assert got[k].reshape(-1).view(torch.uint8).tolist() == v.reshape(-1).view(torch.uint8).tolist() \
            if v.numel() < 10000 else torch.equal(got[k], v), k
    path, _ = _save_tiny(str(tmp_path), "llama")
    sp = SamplingParams(temperature=0, max_tokens=6)
    kw = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=64, max_num_seqs=2)
    outs = []
    for fmt in ("auto", "fastsafetensors", "runai_streamer"):
        llm = LLM(model=path, load_format=fmt, model_loader_extra_config={"concurrency": 3}, **kw)
        outs.append(list(llm.generate(prompt_token_ids=[[5, 6, 7, 8]], sampling_params=sp)[0].outputs[0].token_ids))
        llm.shutdown()
    assert outs[0] == outs[1] == outs[2]


def test_streamed_safetensors_tensors_survive_later_windows(tmp_path):
    """Tensors yielded by the streaming reader are kept by weight loaders (packed 4-bit triples wait for their
    partners, multimodal loaders collect whole sub-models): with windows far smaller than the checkpoint, every tensor
    held until the END of the iteration must still equal the file's contents."""
    import torch
    from safetensors.torch import save_file

    from vllm_distributed_amd.models.loader import streamed_safetensors_iterator
    g = torch.Generator().manual_seed(1)
    want = {f"t{i}": torch.randn(257, 131, generator=g) for i in range(24)}        # 24 x 134 KB, windows of 256 KB
    save_file(want, str(tmp_path / "many.safetensors"))
    held = dict(streamed_safetensors_iterator([str(tmp_path / "many.safetensors")], concurrency=3,
                                              window_bytes=256 << 10))
    assert set(held) == set(want)
    for k, v in want.items():
        assert torch.equal(held[k], v), k
    # views (zero_copy) are only good until the next window: the opt-in exists, and a consumer that copies at once is fine
    got = {k: v.clone() for k, v in streamed_safetensors_iterator([str(tmp_path / "many.safetensors")], concurrency=3,
                                                                  window_bytes=256 << 10, zero_copy=True)}
    for k, v in want.items():
        assert torch.equal(got[k], v), k


def test_tensorizer_loader_round_trip(tmp_path):
    """Serialise a loaded model the way it sits in memory, start a second engine from that one file through the
    native streamed reader with hash verification: same greedy tokens; a flipped byte is caught."""
    import torch
    from tests.test_e2e_cpu import _save_tiny
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.models.loader import TensorizerLoader
    path, _ = _save_tiny(str(tmp_path), "llama")
    kw = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4)
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 20]
    a = LLM(model=path, **kw)
    ref = [list(o.outputs[0].token_ids) for o in a.generate(prompt_token_ids=prompts, sampling_params=sp)]
    model = a.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    uri = str(tmp_path / "ser" / "model-rank-%03d.tensors")
    TensorizerLoader.tensorize_model(model, uri % 0)
    a.shutdown()
    extra = {"tensorizer_uri": uri, "verify_hash": True, "num_readers": 4}
    b = LLM(model=path, load_format="tensorizer", model_loader_extra_config=extra, **kw)
    got = [list(o.outputs[0].token_ids) for o in b.generate(prompt_token_ids=prompts, sampling_params=sp)]
    b.shutdown()
    assert got == ref
    raw = bytearray(open(uri % 0, "rb").read())
    raw[-5] ^= 0x40
    open(uri % 0, "wb").write(bytes(raw))
    with pytest.raises(Exception, match="recorded hash"):
        LLM(model=path, load_format="tensorizer", model_loader_extra_config=extra, **kw)
