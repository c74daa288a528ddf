"""fp8 quantisation plumbing on CPU: online quantisation of a 16-bit checkpoint, loading an
fp8-serialised checkpoint with per-projection scales (fused-layer requantisation), MoE scales."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file, save_file

from tests.test_e2e_cpu import _save_tiny
from vllm_distributed_amd import LLM, SamplingParams

KW = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
          max_num_batched_tokens=64)
PROMPTS = [[5, 17, 99, 3, 250, 11], [7] * 20, [1, 2, 3]]
SP = SamplingParams(temperature=0, max_tokens=6, ignore_eos=True, logprobs=1)


def _first_logprobs(llm):
    outs = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP)
    return [[list(lp.values())[0].logprob for lp in o.outputs[0].logprobs] for o in outs], \
        [list(o.outputs[0].token_ids) for o in outs]


def _to_fp8_checkpoint(path, static_act):
    """Rewrite a tiny HF checkpoint as per-tensor fp8 with one scale per original projection."""
    st = load_file(os.path.join(path, "model.safetensors"))
    out = {}
    for k, v in st.items():
        if k.endswith("proj.weight") or ".experts." in k:
            s = v.abs().max().float() / 448.0
            out[k] = (v.float() / s).to(torch.float8_e4m3fn)
            out[k + "_scale"] = s.reshape(())
            if static_act:
                out[k.replace(".weight", ".input_scale")] = torch.tensor(0.05)
        else:
            out[k] = v
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["quantization_config"] = {"quant_method": "fp8", "activation_scheme": "static" if static_act else "dynamic"}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))


def test_online_fp8_close_to_fp32(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    a = LLM(model=path, **KW)
    ref_lp, ref_tok = _first_logprobs(a)
    a.shutdown()
    b = LLM(model=path, quantization="fp8", **KW)
    m = b.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    assert m.model.layers[0].mlp.gate_up_proj.weight.dtype == torch.float8_e4m3fn
    assert m.lm_head.weight.dtype == torch.float32
    lp, tok = _first_logprobs(b)
    b.shutdown()
    # the first generated token's log-probability moves only by quantisation noise
    for x, y in zip(lp, ref_lp):
        assert abs(x[0] - y[0]) < 0.05


def test_serialized_fp8_checkpoint_matches_dequantised_reference(tmp_path):
    for kind in ("llama", "mixtral"):
        path, _ = _save_tiny(str(tmp_path), kind)
        _to_fp8_checkpoint(path, static_act=(kind == "llama"))
        q = LLM(model=path, **KW)
        lp, tok = _first_logprobs(q)
        q.shutdown()
        # reference: the same fp8 values dequantised back into an ordinary checkpoint
        st = load_file(os.path.join(path, "model.safetensors"))
        deq = {}
        for k, v in st.items():
            if k.endswith("_scale"):
                continue
            deq[k] = v.float() * st[k + "_scale"] if v.dtype == torch.float8_e4m3fn else v
        save_file(deq, os.path.join(path, "model.safetensors"))
        cfg = json.load(open(os.path.join(path, "config.json")))
        cfg.pop("quantization_config")
        json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
        r = LLM(model=path, **KW)
        ref_lp, ref_tok = _first_logprobs(r)
        r.shutdown()
        for x, y in zip(lp, ref_lp):
            assert abs(x[0] - y[0]) < 0.03, kind      # only the fused-shard requantisation differs


def test_fp8_kv_cache_close_to_16bit(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    a = LLM(model=path, **KW)
    ref_lp, ref_tok = _first_logprobs(a)
    a.shutdown()
    b = LLM(model=path, kv_cache_dtype="fp8", **KW)
    layer = next(iter(b.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.attn_layers.values()))
    assert layer.k_cache.dtype == torch.float8_e4m3fn
    lp, tok = _first_logprobs(b)
    b.shutdown()
    for x, y in zip(lp, ref_lp):
        assert abs(x[0] - y[0]) < 0.05


# ---- weight-only formats dequantised at load time --------------------------------------------------
def _pack_rows(v, bits):
    per = 32 // bits
    v = v.to(torch.int64).view(v.shape[0] // per, per, v.shape[1])
    out = torch.zeros(v.shape[0], v.shape[2], dtype=torch.int64)
    for i in range(per):
        out |= v[:, i, :] << (bits * i)
    return out.to(torch.int32) if bits * per < 32 else (out & 0xFFFFFFFF).to(torch.uint32).view(torch.int32)


def _pack_cols(v, bits, order=None):
    per = 32 // bits
    v = v.to(torch.int64).view(v.shape[0], v.shape[1] // per, per)
    if order is not None:
        inv = [order.index(i) for i in range(per)]
        v = v[:, :, inv]
    out = torch.zeros(v.shape[0], v.shape[1], dtype=torch.int64)
    for i in range(per):
        out |= v[:, :, i] << (bits * i)
    return (out & 0xFFFFFFFF).to(torch.uint32).view(torch.int32)


def _int4_quantise(w, group, bits=4):
    """w [N, K] -> q [K, N] in 0..2^bits-1, zeros [G, N], scales [G, N] and the dequantised [N, K]."""
    N, K = w.shape
    top = (1 << bits) - 1
    wt = w.t().reshape(K // group, group, N)
    lo, hi = wt.amin(1), wt.amax(1)
    scales = ((hi - lo) / top).clamp(min=1e-6).half()
    zeros = (-lo / scales.float()).round().clamp(1, top)
    q = (wt / scales.float().unsqueeze(1) + zeros.unsqueeze(1)).round().clamp(0, top)
    deq = ((q - zeros.unsqueeze(1)) * scales.float().unsqueeze(1)).reshape(K, N).t().contiguous()
    return q.reshape(K, N).to(torch.int32), zeros.to(torch.int32), scales, deq


@pytest.mark.parametrize("fmt,packed", [("gptq", True), ("gptq", False), ("awq", True), ("awq", False),
                                        ("fp8_block", False), ("gptq8", True), ("gptq_act", True)])
def test_weight_only_checkpoints_are_dequantised_while_loading(tmp_path, fmt, packed, monkeypatch):
    """A GPTQ / AWQ / block-fp8 checkpoint of the tiny Llama generates the tokens of the HF model
    that holds the dequantised weights, with the 4-bit weights kept packed (W4A16 layers) or expanded
    while loading."""
    import json
    monkeypatch.setenv("VDA_INT4_WEIGHTS", "1" if packed else "0")
    import os

    from safetensors.torch import load_file, save_file

    from tests.test_e2e_cpu import _save_tiny
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.models.dequant import _AWQ_ORDER
    path, hf = _save_tiny(str(tmp_path), "llama")
    sd = load_file(os.path.join(path, "model.safetensors"))
    out, group, orders = {}, 32, {}
    for name, t in sd.items():
        if not (name.endswith("_proj.weight") and t.dim() == 2):
            out[name] = t
            continue
        prefix = name[: -len(".weight")]
        if fmt == "fp8_block":
            bn = bk = 32
            N, K = t.shape
            blocks = t.view(N // bn, bn, K // bk, bk)
            s = (blocks.abs().amax(dim=(1, 3)) / 448.0).clamp(min=1e-8)
            q = (blocks / s[:, None, :, None]).view(N, K).to(torch.float8_e4m3fn)
            out[name], out[prefix + ".weight_scale_inv"] = q, s.float()
            deq = (q.float().view(N // bn, bn, K // bk, bk) * s[:, None, :, None]).view(N, K)
        elif fmt in ("gptq8", "gptq_act"):
            # 8-bit weights, or 4-bit weights quantised in a shuffled column order (desc_act): the projections
            # that share an input share the order, as they do in real files
            bits = 8 if fmt == "gptq8" else 4
            K = t.shape[1]
            order = orders.setdefault(K, torch.randperm(K, generator=torch.Generator().manual_seed(K)))
            if fmt == "gptq8":
                order = torch.arange(K)
            q, zeros, scales, deq = _int4_quantise(t.float()[:, order], group, bits)
            inv = torch.argsort(order)
            q, deq = q[inv], deq[:, inv].contiguous()
            out[prefix + ".qweight"] = _pack_rows(q, bits)
            out[prefix + ".qzeros"] = _pack_cols(zeros - 1, bits)
            out[prefix + ".g_idx"] = (inv // group).to(torch.int32)
            out[prefix + ".scales"] = scales
        else:
            q, zeros, scales, deq = _int4_quantise(t.float(), group)
            if fmt == "gptq":
                out[prefix + ".qweight"] = _pack_rows(q, 4)
                out[prefix + ".qzeros"] = _pack_cols(zeros - 1, 4)          # v1 files store zero - 1
                out[prefix + ".g_idx"] = (torch.arange(q.shape[0]) // group).to(torch.int32)
            else:
                out[prefix + ".qweight"] = _pack_cols(q, 4, _AWQ_ORDER)
                out[prefix + ".qzeros"] = _pack_cols(zeros, 4, _AWQ_ORDER)
            out[prefix + ".scales"] = scales
        mod = hf.get_submodule(prefix)
        with torch.no_grad():
            mod.weight.copy_(deq)
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["quantization_config"] = {"gptq": {"quant_method": "gptq", "bits": 4, "group_size": group, "desc_act": False},
                                  "gptq8": {"quant_method": "gptq", "bits": 8, "group_size": group, "desc_act": False},
                                  "gptq_act": {"quant_method": "gptq", "bits": 4, "group_size": group, "desc_act": True},
                                  "awq": {"quant_method": "awq", "bits": 4, "group_size": group, "zero_point": True,
                                          "version": "gemm"},
                                  "fp8_block": {"quant_method": "fp8", "weight_block_size": [32, 32],
                                                "activation_scheme": "dynamic"}}[fmt]
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64)
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    names = {n for n, _ in llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
             .named_parameters()}
    assert any(n.endswith("qkv_proj.qweight") for n in names) == packed
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (fmt, p)


@pytest.mark.parametrize("strategy", ["channel", "tensor"])
def test_int8_w8a8_checkpoint(tmp_path, strategy):
    """compressed-tensors int8 W8A8 (dynamic per-token activations): the engine's greedy tokens equal a
    plain PyTorch emulation of the same arithmetic through the HF model."""
    path, hf = _save_tiny(str(tmp_path), "llama")
    sd = load_file(os.path.join(path, "model.safetensors"))
    out = {}
    for name, t in sd.items():
        if not (name.endswith("_proj.weight") and t.dim() == 2):
            out[name] = t
            continue
        prefix = name[: -len(".weight")]
        amax = t.abs().amax(1, keepdim=True) if strategy == "channel" else t.abs().amax().reshape(1)
        s = (amax / 127.0).clamp(min=1e-8).float()
        q = torch.round(t / s).clamp(-128, 127).to(torch.int8)
        out[name], out[prefix + ".weight_scale"] = q, s
        mod = hf.get_submodule(prefix)
        wq, ws = q.float(), s.reshape(-1, 1) if strategy == "channel" else s.reshape(1, 1)

        def fwd(x, wq=wq, ws=ws):
            xs = (x.abs().amax(-1, keepdim=True) / 127.0).clamp(min=1e-10)
            xq = torch.round(x / xs).clamp(-128, 127)
            return (xq @ wq.t()) * xs * ws.reshape(1, -1)
        mod.forward = fwd
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["quantization_config"] = {
        "quant_method": "compressed-tensors", "format": "int-quantized", "ignore": ["lm_head"],
        "config_groups": {"group_0": {"targets": ["Linear"],
                                      "weights": {"num_bits": 8, "type": "int", "symmetric": True, "strategy": strategy},
                                      "input_activations": {"num_bits": 8, "type": "int", "symmetric": True,
                                                            "dynamic": True, "strategy": "token"}}}}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64)
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (strategy, p)


@pytest.mark.parametrize("dynamic", [True, False])
def test_int8_w8a8_asymmetric_activations(tmp_path, dynamic):
    """compressed-tensors int8 W8A8 whose activations carry a zero point (dynamic per token, or a static
    scale + zero point from calibration): the correction by azp x column sums of the weight must make the
    engine's greedy tokens equal a plain PyTorch emulation of the same arithmetic through the HF model."""
    path, hf = _save_tiny(str(tmp_path), "llama")
    sd = load_file(os.path.join(path, "model.safetensors"))
    out = {}
    for name, t in sd.items():
        if not (name.endswith("_proj.weight") and t.dim() == 2):
            out[name] = t
            continue
        prefix = name[: -len(".weight")]
        s = (t.abs().amax(1, keepdim=True) / 127.0).clamp(min=1e-8).float()
        q = torch.round(t / s).clamp(-128, 127).to(torch.int8)
        out[name], out[prefix + ".weight_scale"] = q, s
        if not dynamic:
            out[prefix + ".input_scale"] = torch.tensor([0.06])
            out[prefix + ".input_zero_point"] = torch.tensor([-7], dtype=torch.int8)
        mod = hf.get_submodule(prefix)
        wq, ws = q.float(), s.reshape(1, -1)

        def fwd(x, wq=wq, ws=ws):
            if dynamic:
                mx, mn = x.amax(-1, keepdim=True), x.amin(-1, keepdim=True)
                xs = ((mx - mn) / 255.0).clamp(min=1e-10)
                z = torch.round(-128.0 - mn / xs)
            else:
                xs, z = torch.tensor(0.06), torch.tensor(-7.0)
            xq = (torch.round(x / xs) + z).clamp(-128, 127)
            return ((xq - z) @ wq.t()) * xs * ws
        mod.forward = fwd
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["quantization_config"] = {
        "quant_method": "compressed-tensors", "format": "int-quantized", "ignore": ["lm_head"],
        "config_groups": {"group_0": {"targets": ["Linear"],
                                      "weights": {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "channel"},
                                      "input_activations": {"num_bits": 8, "type": "int", "symmetric": False,
                                                            "dynamic": dynamic,
                                                            "strategy": "token" if dynamic else "tensor"}}}}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64)
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (dynamic, p)


_NF4 = [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
        -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
        0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0]


def _bnb4_quantise(w, blocksize=64, nested=True):
    """bitsandbytes-style NF4 with (optionally) double-quantised scales; returns the checkpoint tensors and
    the values they decode to."""
    import json
    qmap = torch.tensor(_NF4)
    flat = w.reshape(-1, blocksize)
    absmax = flat.abs().amax(1).clamp(min=1e-8)
    codes = (flat / absmax[:, None]).unsqueeze(-1).sub(qmap).abs().argmin(-1).to(torch.uint8).view(-1)
    packed = ((codes[0::2] << 4) | codes[1::2]).view(-1, 1)
    state = {"quant_type": "nf4", "blocksize": blocksize, "dtype": "float32", "shape": list(w.shape)}
    out = {"quant_map": qmap}
    if nested:
        nbs = 256
        offset = float(absmax.mean())
        a = absmax - offset
        pad = (-a.numel()) % nbs
        ap = torch.cat([a, a.new_zeros(pad)]).view(-1, nbs)
        n_abs = ap.abs().amax(1).clamp(min=1e-8)
        nmap = torch.linspace(-1, 1, 256)
        ncodes = (ap / n_abs[:, None]).unsqueeze(-1).sub(nmap).abs().argmin(-1).to(torch.uint8).view(-1)[: a.numel()]
        out.update(absmax=ncodes, nested_absmax=n_abs, nested_quant_map=nmap)
        state.update(nested_blocksize=nbs, nested_dtype="float32", nested_offset=offset)
        absmax = nmap[ncodes.long()] * n_abs.repeat_interleave(nbs)[: a.numel()] + offset
    else:
        out["absmax"] = absmax
    out["quant_state.bitsandbytes__nf4"] = torch.tensor(list(json.dumps(state).encode()), dtype=torch.uint8)
    deq = (qmap[codes.long()].view(-1, blocksize) * absmax[:, None]).view(w.shape)
    return packed, out, deq


def _write_bnb_checkpoint(tmp_path, kind):
    import json
    import os

    from safetensors.torch import load_file, save_file

    from tests.test_e2e_cpu import _save_tiny
    path, hf = _save_tiny(str(tmp_path), "llama")
    sd = load_file(os.path.join(path, "model.safetensors"))
    out = {}
    for name, t in sd.items():
        if not (name.endswith("_proj.weight") and t.dim() == 2):
            out[name] = t
            continue
        prefix = name[: -len(".weight")]
        if kind == "int8":
            scb = t.abs().amax(1).clamp(min=1e-8)
            q = (t / scb[:, None] * 127).round().clamp(-127, 127).to(torch.int8)
            out[name], out[prefix + ".SCB"] = q, scb
            out[prefix + ".weight_format"] = torch.tensor(0, dtype=torch.uint8)
            deq = q.float() * scb[:, None] / 127
        else:
            packed, extra, deq = _bnb4_quantise(t.float(), nested=kind == "nf4_nested")
            out[name] = packed
            for k, v in extra.items():
                out[f"{name}.{k}"] = v
        with torch.no_grad():
            hf.get_submodule(prefix).weight.copy_(deq)
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["quantization_config"] = {"quant_method": "bitsandbytes", "load_in_4bit": kind != "int8",
                                  "load_in_8bit": kind == "int8", "bnb_4bit_quant_type": "nf4"}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    return path, hf


@pytest.mark.parametrize("kind", ["nf4_nested", "nf4", "int8"])
def test_bitsandbytes_checkpoint_is_dequantised_while_loading(tmp_path, kind):
    """A pre-quantised bitsandbytes checkpoint (tensor names and packing as bitsandbytes writes them; the
    package itself is not installed, so its own writer is unpinned) decodes to the tokens of the HF model
    holding the dequantised weights."""
    from vllm_distributed_amd import LLM, SamplingParams
    path, hf = _write_bnb_checkpoint(tmp_path, kind)
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64)
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    qkv = model.model.layers[0].self_attn.qkv_proj
    if kind == "int8":
        assert qkv.weight.dtype == torch.float32
    else:
        # the 4-bit codes stay in memory: bytes as the checkpoint holds them, scales decoded to fp32 once
        from vllm_distributed_amd import ops
        assert qkv.qweight.dtype == torch.uint8 and qkv.scales.dtype == torch.float32
        assert qkv.scales.numel() * 64 == qkv.qweight.numel() * 2 and getattr(qkv, "weight", None) is None
        want = torch.cat([hf.model.layers[0].self_attn.get_submodule(n).weight for n in ("q_proj", "k_proj", "v_proj")])
        got = ops.cb4_dequant(qkv.qweight, qkv.scales, qkv.code_table, 64, True, torch.float32,
                              getattr(qkv, "cb4_chunked", False))        # (the GPU keeps a chunk layout)
        torch.testing.assert_close(got.cpu(), want.detach(), rtol=1e-6, atol=1e-8)
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (kind, p)


def _mxfp4_quantise(w: torch.Tensor):
    """-> (packed uint8 [N, K/2], e8m0 scales uint8 [N, K/32], dequantised fp32)."""
    grid = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
    N, K = w.shape
    b = w.float().reshape(N, K // 32, 32)
    e = torch.floor(torch.log2(b.abs().amax(-1, keepdim=True).clamp(min=2.0 ** -100))) - 2
    s = torch.exp2(e)
    v = (b / s).clamp(-6, 6)
    idx = (v.abs().unsqueeze(-1) - grid).abs().argmin(-1)
    codes = (idx + 8 * (v < 0)).to(torch.uint8).reshape(N, K)
    packed = codes[:, 0::2] | (codes[:, 1::2] << 4)
    deq = (grid[idx] * torch.where(v < 0, -1.0, 1.0) * s).reshape(N, K)
    return packed.contiguous(), (e.reshape(N, K // 32) + 127).to(torch.uint8), deq


def test_mxfp4_rounding_of_activations_lands_on_the_grid():
    from vllm_distributed_amd.models.dequant import dequantize_mxfp4, quant_dequant_mxfp4
    torch.manual_seed(0)
    x = torch.randn(6, 64) * 3
    y = quant_dequant_mxfp4(x)
    packed, sc, deq = _mxfp4_quantise(y)                 # already representable: quantising again changes nothing
    assert torch.equal(deq, y)
    assert torch.equal(dequantize_mxfp4(packed, sc, torch.float32), y)
    assert float((y - x).abs().max()) <= float(x.abs().max()) / 4
    assert torch.equal(quant_dequant_mxfp4(torch.zeros(2, 32)), torch.zeros(2, 32))


def test_quark_mxfp4_checkpoint_is_dequantised_while_loading(tmp_path):
    """A quark microscaling-fp4 checkpoint (uint8 `weight` with two codes per byte, uint8 `weight_scale`
    exponents; lm_head excluded) decodes to the tokens of the HF model holding the dequantised weights.
    quark is not installed: its own exporter is unpinned, the layout follows the reference's
    quark_w4a4_mxfp4.py / mxfp4_utils.py."""
    path, hf = _save_tiny(str(tmp_path), "llama")
    sd = load_file(os.path.join(path, "model.safetensors"))
    out = {}
    for name, t in sd.items():
        if not (name.endswith("_proj.weight") and t.dim() == 2):
            out[name] = t
            continue
        prefix = name[: -len(".weight")]
        out[name], out[prefix + ".weight_scale"], deq = _mxfp4_quantise(t.float())
        with torch.no_grad():
            hf.get_submodule(prefix).weight.copy_(deq)
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    spec = {"dtype": "fp4", "group_size": 32, "scale_format": "e8m0", "qscheme": "per_group", "is_dynamic": False}
    cfg["quantization_config"] = {"quant_method": "quark", "exclude": ["lm_head"],
                                  "global_quant_config": {"weight": spec, "input_tensors": dict(spec, is_dynamic=True)}}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64)
    # linear layers keep the codes as 8-bit integers with power-of-two group scales: exactly the same values
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    qkv = model.model.layers[0].self_attn.qkv_proj
    assert qkv.qweight.dtype == torch.uint8 and qkv.scales.shape[1] == qkv.qweight.shape[1] // 32
    assert not hasattr(qkv, "weight") or qkv.weight is None
    want = torch.cat([hf.model.layers[0].self_attn.get_submodule(n).weight for n in ("q_proj", "k_proj", "v_proj")])
    got = (qkv.qweight.float() - qkv.qzeros.repeat_interleave(32, 1)) * qkv.scales.repeat_interleave(32, 1)
    assert torch.equal(got.cpu(), want.detach())
    assert model.lm_head.weight.dtype == torch.float32                      # excluded layers stay as they are
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, p


def test_modelopt_nvfp4_checkpoint_is_dequantised_while_loading(tmp_path):
    """ModelOpt NVFP4 layout (uint8 `weight`, fp8 `weight_scale` per 16, fp32 `weight_scale_2` and
    `input_scale`; names as the reference's modelopt.py registers them, the exporter itself is not
    installed) decodes to the tokens of the HF model holding the dequantised weights."""
    grid = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
    path, hf = _save_tiny(str(tmp_path), "llama")
    sd = load_file(os.path.join(path, "model.safetensors"))
    out = {}
    for name, t in sd.items():
        if not (name.endswith("_proj.weight") and t.dim() == 2):
            out[name] = t
            continue
        prefix = name[: -len(".weight")]
        N, K = t.shape
        g2 = t.abs().max().float() / (6.0 * 448.0)
        b = t.float().reshape(N, K // 16, 16)
        sc = (b.abs().amax(-1) / 6.0 / g2).clamp(min=2.0 ** -9).to(torch.float8_e4m3fn)
        s = sc.float().unsqueeze(-1) * g2
        v = (b / s).clamp(-6, 6)
        idx = (v.abs().unsqueeze(-1) - grid).abs().argmin(-1)
        codes = (idx + 8 * (v < 0)).to(torch.uint8).reshape(N, K)
        out[name] = (codes[:, 0::2] | (codes[:, 1::2] << 4)).contiguous()
        out[prefix + ".weight_scale"] = sc
        out[prefix + ".weight_scale_2"] = g2.reshape(())
        out[prefix + ".input_scale"] = torch.tensor(0.01)
        with torch.no_grad():
            hf.get_submodule(prefix).weight.copy_((grid[idx] * torch.where(v < 0, -1.0, 1.0) * s).reshape(N, K))
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["quantization_config"] = {"quant_method": "modelopt", "quant_algo": "NVFP4", "group_size": 16,
                                  "kv_cache_quant_algo": None, "exclude_modules": ["lm_head"]}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64)
    # linear layers keep the E2M1 codes; group scales carry the per-tensor scale of their projection
    from vllm_distributed_amd import ops
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    qkv = model.model.layers[0].self_attn.qkv_proj
    assert qkv.qweight.dtype == torch.uint8 and qkv.scales.shape[1] * 16 == qkv.qweight.shape[1] * 2
    want = torch.cat([hf.model.layers[0].self_attn.get_submodule(n).weight for n in ("q_proj", "k_proj", "v_proj")])
    got = ops.cb4_dequant(qkv.qweight, qkv.scales, qkv.code_table, 16, False, torch.float32)
    torch.testing.assert_close(got.cpu(), want.detach(), rtol=1e-6, atol=1e-9)
    assert model.lm_head.weight.dtype == torch.float32
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, p


@pytest.mark.parametrize("kind", ["llama", "mixtral"])
def test_block_scaled_fp8_checkpoint_runs_natively(tmp_path, kind, monkeypatch):
    """128 x 128 block-scaled fp8 (DeepSeek-V3 style ``weight_scale_inv``): the weights stay e4m3 in the
    model (half the bytes of the expanded form), activations are quantised per group of 128 on the fly, and
    the tokens equal those of the HF model holding the dequantised weights with the same activation rounding
    emulated in front of every quantised projection."""
    import json
    import os

    import transformers as tf
    from safetensors.torch import load_file, save_file

    from vllm_distributed_amd import LLM, SamplingParams
    torch.manual_seed(0)
    if kind == "llama":
        cfg = tf.LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=256, num_hidden_layers=2,
                             num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256, head_dim=64)
        hf = tf.LlamaForCausalLM(cfg)
    else:
        cfg = tf.MixtralConfig(vocab_size=320, hidden_size=128, intermediate_size=128, num_hidden_layers=2,
                               num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=256,
                               num_local_experts=4, num_experts_per_tok=2, sliding_window=None, head_dim=64)
        hf = tf.MixtralForCausalLM(cfg)
    hf = hf.float().eval()
    path = str(tmp_path / kind)
    hf.save_pretrained(path, safe_serialization=True)
    sd = load_file(os.path.join(path, "model.safetensors"))
    out, n_q, emulated_all, deq_sd = {}, 0, True, {}

    def _round128(x):
        from vllm_distributed_amd.ops import reference as R
        xq, xs = R.per_group_fp8_quant(x.reshape(-1, x.shape[-1]), 128)
        return (xq.float().view(*xs.shape, -1) * xs.unsqueeze(-1)).view(x.shape)
    for name, t in sd.items():
        leaf = name.rsplit(".", 2)[-2]
        if not (t.dim() == 2 and name.endswith(".weight") and (leaf.endswith("_proj") or leaf in ("w1", "w2", "w3"))):
            out[name] = t
            continue
        N, K = t.shape
        bn = bk = 128
        blocks = t.view(N // bn, bn, K // bk, bk)
        s = (blocks.abs().amax(dim=(1, 3)) / 448.0).clamp(min=1e-8)
        q = (blocks / s[:, None, :, None]).view(N, K).to(torch.float8_e4m3fn)
        out[name], out[name[: -len(".weight")] + ".weight_scale_inv"] = q, s.float()
        deq = (q.float().view(N // bn, bn, K // bk, bk) * s[:, None, :, None]).view(N, K)
        n_q += 1
        try:
            mod = hf.get_submodule(name[: -len(".weight")])
        except AttributeError:
            # experts held as stacked tensors by this transformers version: write the dequantised weights
            # back through a reload below, without the activation rounding (loose comparison for them)
            emulated_all = False
            deq_sd[name] = deq
            continue
        with torch.no_grad():
            mod.weight.copy_(deq)

        def rounded(x, _m=mod):
            from vllm_distributed_amd.ops import reference as R
            xq, xs = R.per_group_fp8_quant(x.reshape(-1, x.shape[-1]), 128)
            xd = (xq.float().view(*xs.shape, -1) * xs.unsqueeze(-1)).view(x.shape)
            return torch.nn.functional.linear(xd, _m.weight, _m.bias)
        mod.forward = rounded
    assert n_q > 0
    if deq_sd:
        full = dict(sd)
        full.update(deq_sd)
        save_file(full, os.path.join(path, "model.safetensors"))
        fresh = type(hf).from_pretrained(path).float().eval()
        for n_, m_ in hf.named_modules():            # keep the patched forwards of the attention projections
            if "forward" in m_.__dict__:
                tgt = fresh.get_submodule(n_)
                with torch.no_grad():
                    tgt.weight.copy_(m_.weight)
                fwd = m_.__dict__["forward"]
                tgt.forward = (lambda x, _f=fwd.__defaults__, _m=tgt: __import__("torch").nn.functional.linear(
                    _round128(x), _m.weight, _m.bias))
        hf = fresh
    save_file(out, os.path.join(path, "model.safetensors"))
    c = json.load(open(os.path.join(path, "config.json")))
    c["quantization_config"] = {"quant_method": "fp8", "weight_block_size": [128, 128], "activation_scheme": "dynamic"}
    json.dump(c, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30, list(range(40, 75))]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=8,
                                                                                 ignore_eos=True))
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    params = dict(model.named_parameters())
    fp8 = [n for n, p in params.items() if p.dtype == torch.float8_e4m3fn]
    assert fp8 and any(n.endswith("weight_scale_inv") for n in params), "weights must stay in fp8 with block scales"
    llm.shutdown()
    same = total = 0
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        got = list(o.outputs[0].token_ids)
        same += sum(a == b for a, b in zip(got, ref))
        total += len(ref)
        if emulated_all:
            assert got[:4] == ref[:4], (kind, p, got, ref)
        else:
            assert got[0] == ref[0], (kind, p, got, ref)
    # fp32 accumulation order differs: allow a late near-tie (more of them where the reference does not
    # round the experts' activations)
    assert same >= (0.9 if emulated_all else 0.6) * total, (same, total)


def test_bitsandbytes_in_flight_nf4(tmp_path):
    """quantization="bitsandbytes" on a 16-bit checkpoint: weights become NF4 while loading (packed bytes + one
    absmax per 64 values in memory); greedy tokens equal the HF model whose projections were replaced by the
    NF4-rounded weights; the code book round-trips its own values exactly."""
    from vllm_distributed_amd.layers.quant import NF4_CODE, nf4_dequantize, nf4_quantize
    code = torch.tensor(NF4_CODE)
    w = (code[torch.randint(0, 16, (8, 128))] * 0.37)
    w[:, ::64] = 0.37                                   # every block's absmax is the same scale
    p, a = nf4_quantize(w)
    assert p.shape == (8, 64) and a.shape == (8, 2)
    torch.testing.assert_close(nf4_dequantize(p, a, torch.float32), w, rtol=1e-6, atol=1e-7)
    path, hf = _save_tiny(str(tmp_path), "llama")
    with torch.no_grad():
        for name, mod in hf.named_modules():
            if isinstance(mod, torch.nn.Linear) and name != "lm_head":
                mod.weight.copy_(nf4_dequantize(*nf4_quantize(mod.weight.data), torch.float32))
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64, quantization="bitsandbytes")
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    qkv = model.model.layers[0].self_attn.qkv_proj
    assert qkv.qweight.dtype == torch.uint8 and qkv.qweight.shape[1] * 2 == 128
    assert qkv.scales.shape[0 if getattr(qkv, "cb4_chunked", False) else 1] == 2
    assert getattr(qkv, "weight", None) is None and qkv.code_table.tolist() == list(code.tolist())
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    llm.shutdown()
    for pr, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([pr]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(pr):].tolist()
        assert list(o.outputs[0].token_ids) == ref


def test_code_book_layers_shard_over_two_ranks(tmp_path):
    """Tensor parallel 2 (two worker processes over gloo) on a pre-quantised NF4 checkpoint with double-quantised
    scales: column-parallel layers take their rows of codes and scales, row-parallel layers their byte and group
    columns; the tokens are transformers' for the dequantised weights."""
    from vllm_distributed_amd import LLM, SamplingParams
    path, hf = _write_bnb_checkpoint(tmp_path, "nf4_nested")
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              max_num_batched_tokens=64, distributed_executor_backend="mp", tensor_parallel_size=2)
    try:
        outs = llm.generate(prompt_token_ids=prompts,
                            sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    finally:
        llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, p
