"""GGUF: container parsing, block dequantisers against scalar re-implementations of the ggml
layouts written here, and an end-to-end run of a tiny Llama exported to GGUF by the writer below
(llama.cpp's q/k row permutation included).  The `gguf` package is not installed, so parity with
llama.cpp's own writer is unpinned; the layouts follow ggml-quants.c."""
import os
import struct

import numpy as np
import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.models import gguf as G


# ---- a minimal writer ------------------------------------------------------------------------------------
def _s(x: str) -> bytes:
    b = x.encode()
    return struct.pack("<Q", len(b)) + b


def _kv(key, val) -> bytes:
    out = _s(key)
    if isinstance(val, str):
        return out + struct.pack("<I", 8) + _s(val)
    if isinstance(val, float):
        return out + struct.pack("<If", 6, val)
    if isinstance(val, int):
        return out + struct.pack("<II", 4, val)
    if isinstance(val, list):          # list of strings
        return out + struct.pack("<IIQ", 9, 8, len(val)) + b"".join(_s(v) for v in val)
    raise TypeError(val)


def write_gguf(path, metadata: dict, tensors: list):
    """tensors: (name, row-major shape, ggml type id, raw bytes)."""
    head = b"GGUF" + struct.pack("<IQQ", 3, len(tensors), len(metadata))
    head += b"".join(_kv(k, v) for k, v in metadata.items())
    off, infos, blobs = 0, b"", []
    for name, shape, t, raw in tensors:
        infos += _s(name) + struct.pack("<I", len(shape)) + b"".join(struct.pack("<Q", d) for d in reversed(shape))
        infos += struct.pack("<IQ", t, off)
        pad = (-len(raw)) % 32
        blobs.append(raw + b"\0" * pad)
        off += len(raw) + pad
    head += infos
    head += b"\0" * ((-len(head)) % 32)
    with open(path, "wb") as f:
        f.write(head + b"".join(blobs))


def quant_q8_0(w: np.ndarray) -> bytes:
    b = w.reshape(-1, 32).astype(np.float32)
    d = np.abs(b).max(1, keepdims=True) / 127.0
    q = np.where(d > 0, np.round(b / np.where(d > 0, d, 1)), 0).astype(np.int8)
    return b"".join(d[i].astype(np.float16).tobytes() + q[i].tobytes() for i in range(b.shape[0]))


def quant_q4_0(w: np.ndarray) -> bytes:
    b = w.reshape(-1, 32).astype(np.float32)
    idx = np.abs(b).argmax(1)
    d = b[np.arange(b.shape[0]), idx] / -8.0
    q = np.clip(np.round(b / np.where(d != 0, d, 1)[:, None]) + 8, 0, 15).astype(np.uint8)
    qs = q[:, :16] | (q[:, 16:] << 4)
    return b"".join(d[i].astype(np.float16).tobytes() + qs[i].tobytes() for i in range(b.shape[0]))


def quant_q4_1(w: np.ndarray) -> bytes:
    b = w.reshape(-1, 32).astype(np.float32)
    lo, hi = b.min(1), b.max(1)
    d = (hi - lo) / 15.0
    q = np.clip(np.round((b - lo[:, None]) / np.where(d != 0, d, 1)[:, None]), 0, 15).astype(np.uint8)
    qs = q[:, :16] | (q[:, 16:] << 4)
    return b"".join(d[i].astype(np.float16).tobytes() + lo[i].astype(np.float16).tobytes() + qs[i].tobytes()
                    for i in range(b.shape[0]))


# ---- scalar references (ggml-quants.c, one block at a time) ------------------------------------------------
def _h(b, o):
    return float(np.frombuffer(bytes(b[o:o + 2]), dtype=np.float16)[0])


def ref_q4_0(b):
    d = _h(b, 0)
    return [d * ((b[2 + j] & 15) - 8) for j in range(16)] + [d * ((b[2 + j] >> 4) - 8) for j in range(16)]


def ref_q5_1(b):
    d, m = _h(b, 0), _h(b, 2)
    qh = int.from_bytes(bytes(b[4:8]), "little")
    lo = [d * ((b[8 + j] & 15) | (((qh >> j) & 1) << 4)) + m for j in range(16)]
    hi = [d * ((b[8 + j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) + m for j in range(16)]
    return lo + hi


def _scale_min(j, q):
    if j < 4:
        return q[j] & 63, q[j + 4] & 63
    return (q[j + 4] & 15) | ((q[j - 4] >> 6) << 4), (q[j + 4] >> 4) | ((q[j] >> 6) << 4)


def ref_q4_k(b):
    d, dmin = _h(b, 0), _h(b, 2)
    sc, q, out, i = b[4:16], b[16:], [], 0
    for c in range(4):
        s1, m1 = _scale_min(i, sc)
        s2, m2 = _scale_min(i + 1, sc)
        out += [d * s1 * (q[32 * c + l] & 15) - dmin * m1 for l in range(32)]
        out += [d * s2 * (q[32 * c + l] >> 4) - dmin * m2 for l in range(32)]
        i += 2
    return out


def ref_q5_k(b):
    d, dmin = _h(b, 0), _h(b, 2)
    sc, qh, q, out, i = b[4:16], b[16:48], b[48:], [], 0
    for c in range(4):
        s1, m1 = _scale_min(i, sc)
        s2, m2 = _scale_min(i + 1, sc)
        out += [d * s1 * ((q[32 * c + l] & 15) + (16 if qh[l] & (1 << (2 * c)) else 0)) - dmin * m1 for l in range(32)]
        out += [d * s2 * ((q[32 * c + l] >> 4) + (16 if qh[l] & (2 << (2 * c)) else 0)) - dmin * m2 for l in range(32)]
        i += 2
    return out


def ref_q6_k(b):
    ql, qh, sc, d = b[:128], b[128:192], np.frombuffer(bytes(b[192:208]), dtype=np.int8), _h(b, 208)
    out = [0.0] * 256
    for n in range(2):
        for l in range(32):
            i = l // 16
            h = qh[32 * n + l]
            q1 = ((ql[64 * n + l] & 15) | ((h & 3) << 4)) - 32
            q2 = ((ql[64 * n + l + 32] & 15) | (((h >> 2) & 3) << 4)) - 32
            q3 = ((ql[64 * n + l] >> 4) | (((h >> 4) & 3) << 4)) - 32
            q4 = ((ql[64 * n + l + 32] >> 4) | (((h >> 6) & 3) << 4)) - 32
            for k, qv in enumerate((q1, q2, q3, q4)):
                out[128 * n + 32 * k + l] = d * float(sc[8 * n + i + 2 * k]) * qv
    return out


@pytest.mark.parametrize("tid,ref", [(2, ref_q4_0), (7, ref_q5_1), (12, ref_q4_k), (13, ref_q5_k), (14, ref_q6_k)])
def test_block_dequantisers_match_scalar_reference(tid, ref):
    name, per, nbytes = G.GGML_TYPES[tid]
    rng = np.random.default_rng(tid)
    blocks = rng.integers(0, 256, size=(5, nbytes), dtype=np.uint8)
    # keep the fp16 scale fields finite and small
    for o in ((0, 2) if name != "Q6_K" else (208,)):
        if name in ("Q4_0",) and o == 2:
            continue
        blocks[:, o:o + 2] = np.frombuffer(rng.uniform(0.01, 0.5, 5).astype(np.float16).tobytes(),
                                           dtype=np.uint8).reshape(5, 2)
    got = G.dequantize(blocks.reshape(-1), tid, (5, per)).numpy()
    want = np.array([ref([int(x) for x in blocks[i]]) for i in range(5)], dtype=np.float32)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_q5_0_and_q8_0_roundtrip():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((4, 64)).astype(np.float32)
    raw = np.frombuffer(quant_q8_0(w), dtype=np.uint8)
    got = G.dequantize(raw, 8, (4, 64)).numpy()
    assert np.abs(got - w).max() < np.abs(w).max() / 100
    # Q5_0: a block whose 5-bit codes are known
    codes = np.arange(32, dtype=np.uint8)
    qs = (codes[:16] & 15) | ((codes[16:] & 15) << 4)
    qh = sum(int(codes[j] >> 4) << j for j in range(32))
    blk = np.float16(0.5).tobytes() + qh.to_bytes(4, "little") + qs.tobytes()
    got = G.dequantize(np.frombuffer(blk, dtype=np.uint8), 6, (32,)).numpy()
    np.testing.assert_allclose(got, 0.5 * (codes.astype(np.float32) - 16))


@pytest.mark.parametrize("tid", [2, 3, 6, 7, 8, 12, 13])
def test_packed_codes_reproduce_the_block_dequantisers(tid):
    """Codes, scales and zero points kept for the packed-weight kernels expand to the values of the float
    dequantiser of the same type, blocks with a zero scale included."""
    from vllm_distributed_amd import ops
    name, per, nbytes = G.GGML_TYPES[tid]
    rng = np.random.default_rng(tid)
    N, K = 6, 512
    blocks = rng.integers(0, 256, size=(N * K // per, nbytes), dtype=np.uint8)
    fields = {"Q4_0": (0,), "Q5_0": (0,), "Q8_0": (0,)}.get(name, (0, 2))
    for o in fields:
        vals = rng.uniform(0.01, 0.5, blocks.shape[0]).astype(np.float16)
        vals[0] = 0                                   # a block whose scale (or minimum) is exactly zero
        blocks[:, o:o + 2] = np.frombuffer(vals.tobytes(), dtype=np.uint8).reshape(-1, 2)
    want = G.dequantize(blocks.reshape(-1), tid, (N, K))
    for bits in ((4, 8) if G._BLOCKS[tid][0] == 4 else (8,)):
        t = G.packed_tensors(blocks.reshape(-1), tid, (N, K), bits, torch.float32)
        assert t["qweight"].shape == ((N, K // 8) if bits == 4 else (N, K)) and t["scales"].shape == (N, K // 32)
        got = ops.wq_dequant(t["qweight"], t["scales"], t["qzeros"], 32, bits, torch.float32)
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


def test_packed_config_follows_the_fused_layers(tmp_path):
    """A fused layer is packed only when every projection in it is: q and k in Q4_0 with v in Q6_K stay 16 bit,
    gate in Q4_0 with up in Q8_0 becomes an 8-bit layer."""
    path = str(tmp_path / "mix.gguf")
    w = np.ones((64, 256), np.float32)
    q6 = bytes(210 * (64 * 256 // 256))
    tensors = [("blk.0.attn_q.weight", w.shape, 2, quant_q4_0(w)), ("blk.0.attn_k.weight", w.shape, 2, quant_q4_0(w)),
               ("blk.0.attn_v.weight", w.shape, 14, q6), ("blk.0.attn_output.weight", w.shape, 3, quant_q4_1(w)),
               ("blk.0.ffn_gate.weight", w.shape, 2, quant_q4_0(w)), ("blk.0.ffn_up.weight", w.shape, 8, quant_q8_0(w)),
               ("blk.0.ffn_down.weight", w.shape, 14, q6), ("token_embd.weight", w.shape, 8, quant_q8_0(w))]
    write_gguf(path, {"general.architecture": "llama", "llama.attention.head_count": 2}, tensors)
    cfg = G.GGUFPackedConfig(path)
    L = "model.layers.0."
    assert cfg.bits_for(L + "self_attn.qkv_proj") is None and cfg.bits_for(L + "self_attn.q_proj") is None
    assert cfg.bits_for(L + "self_attn.o_proj") == 4
    assert cfg.bits_for(L + "mlp.gate_up_proj") == 8 and cfg.bits_for(L + "mlp.gate_proj") == 8
    assert cfg.bits_for(L + "mlp.down_proj") is None and cfg.bits_for("model.embed_tokens") is None
    names = [n for n, _ in G.gguf_weights_iterator(path, torch.float16, cfg)]
    assert L + "self_attn.q_proj.weight" in names and L + "self_attn.o_proj.qweight" in names
    assert L + "mlp.up_proj.qzeros" in names and L + "mlp.down_proj.weight" in names


def _export(hf, path, quant):
    c = hf.config
    H, KV = c.num_attention_heads, c.num_key_value_heads
    md = {"general.architecture": "llama", "general.alignment": 32, "llama.block_count": c.num_hidden_layers,
          "llama.embedding_length": c.hidden_size, "llama.feed_forward_length": c.intermediate_size,
          "llama.attention.head_count": H, "llama.attention.head_count_kv": KV,
          "llama.attention.layer_norm_rms_epsilon": float(c.rms_norm_eps), "llama.rope.freq_base": 10000.0,
          "llama.context_length": 256, "llama.vocab_size": c.vocab_size, "llama.attention.key_length": c.head_dim,
          "tokenizer.ggml.tokens": [f"t{i}" for i in range(4)]}
    names = {"input_layernorm": "attn_norm", "self_attn.q_proj": "attn_q", "self_attn.k_proj": "attn_k",
             "self_attn.v_proj": "attn_v", "self_attn.o_proj": "attn_output", "post_attention_layernorm": "ffn_norm",
             "mlp.gate_proj": "ffn_gate", "mlp.up_proj": "ffn_up", "mlp.down_proj": "ffn_down"}

    def permute(w, n):
        return w.reshape(n, 2, w.shape[0] // n // 2, *w.shape[1:]).swapaxes(1, 2).reshape(w.shape)

    tensors = []
    for k, v in hf.state_dict().items():
        w = v.numpy()
        if k == "model.embed_tokens.weight":
            n = "token_embd.weight"
        elif k == "model.norm.weight":
            n = "output_norm.weight"
        elif k == "lm_head.weight":
            n = "output.weight"
        else:
            _, _, i, rest = k.split(".", 3)
            mod, suffix = rest.rsplit(".", 1)
            n = f"blk.{i}.{names[mod]}.{suffix}"
            if mod == "self_attn.q_proj":
                w = permute(w, H)
            elif mod == "self_attn.k_proj":
                w = permute(w, KV)
        if w.ndim == 1:
            tensors.append((n, w.shape, 0, w.astype(np.float32).tobytes()))
        elif quant:
            # "mixed": 4-bit attention and down projections, the 8-bit type elsewhere (a fused 8-bit gate / up layer)
            kind = quant if quant != "mixed" else ("q4_1" if "attn" in n else "q4_0" if "ffn_down" in n else "q8_0")
            if not n.startswith("blk."):
                kind = "q8_0"
            tid, fn = {"q8_0": (8, quant_q8_0), "q4_0": (2, quant_q4_0), "q4_1": (3, quant_q4_1)}[kind]
            tensors.append((n, w.shape, tid, fn(w)))
        else:
            tensors.append((n, w.shape, 1, w.astype(np.float16).tobytes()))
    tensors.append(("rope_freqs.weight", (c.head_dim // 2,), 0, np.ones(c.head_dim // 2, np.float32).tobytes()))
    write_gguf(path, md, tensors)


@pytest.mark.parametrize("quant", [False, "q8_0", "mixed", "expanded"])
def test_gguf_llama_end_to_end(tmp_path, quant, monkeypatch):
    packed = quant not in (False, "expanded")
    if quant == "expanded":
        monkeypatch.setenv("VDA_GGUF_PACKED", "0")
        quant = "q8_0"
    from tests.test_e2e_cpu import _save_tiny
    _, hf = _save_tiny(str(tmp_path), "llama")
    hf = hf.half().float()             # the F16 export is then exact
    path = str(tmp_path / "tiny.gguf")
    _export(hf, path, quant)
    cfg = G.config_from_gguf(path)
    assert cfg["num_key_value_heads"] == 1 and cfg.get("head_dim", 64) == 64 and cfg["tie_word_embeddings"] is False
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=10,
                                                                                   ignore_eos=True, logprobs=0))
    names = {n for n, _ in llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
             .named_parameters()}
    assert any(n.endswith("qkv_proj.qweight") for n in names) == packed       # the block codes stay packed
    if quant:
        # compare against the model that holds the dequantised weights: the packed layers must match it closely
        r = G.GGUFReader(path)
        sd = hf.state_dict()
        for name, info in r.tensors.items():
            target = G.hf_name(name)
            if target is None or len(info.shape) != 2:
                continue
            w = r.load(name, torch.float32)
            if name.endswith((".attn_q.weight", ".attn_k.weight")):
                w = G._unpermute(w, hf.config.num_attention_heads if ".attn_q." in name else hf.config.num_key_value_heads)
            sd[target].copy_(w)
        r.close()
    for p, o in zip(prompts, outs):
        gen = list(o.outputs[0].token_ids)
        with torch.no_grad():
            lp = torch.log_softmax(hf(torch.tensor([p + gen])).logits[0].float(), -1)
        if not quant:
            ref = hf.generate(torch.tensor([p]), max_new_tokens=10, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
            assert gen == ref
        # teacher-forced: the reported log-probabilities follow the unquantised model's
        worst = max(abs(float(lp[len(p) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob)
                    for k, t in enumerate(gen))
        assert worst < 1e-3, worst
    llm.shutdown()


def test_truncated_file_is_rejected(tmp_path):
    path = str(tmp_path / "bad.gguf")
    write_gguf(path, {"general.architecture": "llama"}, [("token_embd.weight", (4, 32), 8, quant_q8_0(np.ones((4, 32))))])
    data = open(path, "rb").read()
    open(path, "wb").write(data[:-40])
    r = G.GGUFReader(path)
    with pytest.raises(ValueError):
        r.raw(r.tensors["token_embd.weight"])
    r.close()
