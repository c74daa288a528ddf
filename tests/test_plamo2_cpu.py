"""PLaMo 2: no transformers class exists (parity unpinned).  The engine's greedy tokens are compared with the network
written out in fp32, the state-space recurrence as a plain loop over time, prompts prefilled in chunks so that the
convolution window and the scan state cross step boundaries."""
import json
import os

import torch
import torch.nn.functional as F
from safetensors.torch import save_file

from vllm_distributed_amd import LLM, SamplingParams

D, HD, MH, NS, KC, RANK, L, H, KV, FF, V = 64, 16, 8, 16, 4, 64, 4, 4, 2, 96, 320
CFG = dict(architectures=["Plamo2ForCausalLM"], model_type="plamo2", vocab_size=V, hidden_size=D, num_hidden_layers=L,
           hidden_size_per_head=HD, mamba_num_heads=MH, mamba_d_state=NS, mamba_d_conv=KC, mamba_step=2,
           mamba_enabled=True, num_attention_heads=H, num_key_value_heads=KV, intermediate_size=FF, rms_norm_eps=1e-6,
           max_position_embeddings=256, rope_theta=10000.0, tie_word_embeddings=True, torch_dtype="float32")
OFFS = {"pre_mixer_norm": 1.0, "post_mixer_norm": 0.2, "pre_mlp_norm": 1.0, "post_mlp_norm": 1.0 / 5 ** 1.5}


def _weights():
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g) * 0.2                       # noqa: E731
    sd = {"model.embed_tokens.weight": r(V, D), "model.norm.weight": r(D) * 0.1}
    for i in range(L):
        p = f"model.layers.layers.{i}."
        for n in OFFS:
            sd[p + n + ".weight"] = r(D) * 0.1
        sd[p + "mlp.gate_up_proj.weight"], sd[p + "mlp.down_proj.weight"] = r(2 * FF, D), r(D, FF)
        if i % 2 != 1:
            sd.update({p + "mixer.in_proj.weight": r(2 * MH * HD, D), p + "mixer.conv1d.weight": r(MH * HD, 1, KC),
                       p + "mixer.bcdt_proj.weight": r(2 * NS + RANK, MH * HD), p + "mixer.dt_proj.weight": r(MH, RANK),
                       p + "mixer.dt_bias": r(MH), p + "mixer.A_log": r(MH), p + "mixer.D": 1 + r(MH),
                       p + "mixer.B_norm_weight": 1 + r(NS) * 0.1, p + "mixer.C_norm_weight": 1 + r(NS) * 0.1,
                       p + "mixer.dt_norm_weight": 1 + r(RANK) * 0.1, p + "mixer.out_proj.weight": r(D, MH * HD)})
        else:
            sd.update({p + "mixer.qkv_proj.weight": r((H + 2 * KV) * HD, D), p + "mixer.o_proj.weight": r(D, H * HD),
                       p + "mixer.q_weight": 1 + r(H, HD) * 0.1, p + "mixer.k_weight": 1 + r(KV, HD) * 0.1})
    return sd


def _rms(x, w, eps=1e-6):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def _dense_forward(sd, ids):
    T = len(ids)
    x = sd["model.embed_tokens.weight"][ids]
    inv = 1.0 / 10000.0 ** (torch.arange(0, HD, 2).float() / HD)
    ang = torch.arange(T).float()[:, None] * inv[None]
    cos, sin = torch.cat([ang.cos()] * 2, -1), torch.cat([ang.sin()] * 2, -1)

    def rope(t):
        a, b = t[..., : HD // 2], t[..., HD // 2:]
        return t * cos[:, None] + torch.cat([-b, a], -1) * sin[:, None]
    for i in range(L):
        p = f"model.layers.layers.{i}."
        w = lambda n: sd[p + n]                                               # noqa: E731
        h = _rms(x, w("pre_mixer_norm.weight") + 1.0)
        if i % 2 != 1:
            proj = (h @ w("mixer.in_proj.weight").t()).view(T, MH, 2 * HD)
            gate, u = proj[..., :HD].reshape(T, -1), proj[..., HD:].reshape(T, -1)
            pad = F.pad(u.t()[None], (KC - 1, 0))
            u = F.silu(F.conv1d(pad, w("mixer.conv1d.weight"), groups=MH * HD))[0].t()      # [T, MH * HD]
            bcdt = u @ w("mixer.bcdt_proj.weight").t()
            B, C = _rms(bcdt[:, :NS], w("mixer.B_norm_weight")), _rms(bcdt[:, NS:2 * NS], w("mixer.C_norm_weight"))
            dt = _rms(bcdt[:, 2 * NS:], w("mixer.dt_norm_weight")) @ w("mixer.dt_proj.weight").t()
            dt = F.softplus(dt + w("mixer.dt_bias"))                                          # [T, MH]
            A = -torch.exp(w("mixer.A_log"))
            state = torch.zeros(MH, HD, NS)
            ys = []
            uh = u.view(T, MH, HD)
            for t in range(T):
                state = state * torch.exp(dt[t] * A)[:, None, None] + \
                    (dt[t][:, None] * uh[t])[..., None] * B[t][None, None]
                ys.append((state * C[t][None, None]).sum(-1) + w("mixer.D")[:, None] * uh[t])
            y = torch.stack(ys).reshape(T, -1) * F.silu(gate)
            m = y @ w("mixer.out_proj.weight").t()
        else:
            qkv = h @ w("mixer.qkv_proj.weight").t()
            q, k, v = qkv.split([H * HD, KV * HD, KV * HD], -1)
            q = _rms(q.view(T, H, HD), w("mixer.q_weight"))
            k = _rms(k.view(T, KV, HD), w("mixer.k_weight"))
            q, k = rope(q), rope(k)
            k, v = k.repeat_interleave(H // KV, 1), v.view(T, KV, HD).repeat_interleave(H // KV, 1)
            sc = torch.einsum("thd,shd->hts", q, k) * HD ** -0.5
            sc = sc.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float("-inf"))
            m = torch.einsum("hts,shd->thd", sc.softmax(-1), v).reshape(T, -1) @ w("mixer.o_proj.weight").t()
        x = x + _rms(m, w("post_mixer_norm.weight") + OFFS["post_mixer_norm"])
        h = _rms(x, w("pre_mlp_norm.weight") + 1.0)
        gu = h @ w("mlp.gate_up_proj.weight").t()
        m = (F.silu(gu[:, :FF]) * gu[:, FF:]) @ w("mlp.down_proj.weight").t()
        x = x + _rms(m, w("post_mlp_norm.weight") + OFFS["post_mlp_norm"])
    return _rms(x[-1], sd["model.norm.weight"] + 1.0) @ sd["model.embed_tokens.weight"].t()


def test_layer_pattern():
    from types import SimpleNamespace as NS_
    from vllm_distributed_amd.models.plamo2 import is_mamba_layer
    c = NS_(mamba_enabled=True, mamba_step=4, num_hidden_layers=8)
    assert [is_mamba_layer(c, i) for i in range(8)] == [True, True, False, True] * 2
    c = NS_(mamba_enabled=True, mamba_step=8, num_hidden_layers=3)
    assert [is_mamba_layer(c, i) for i in range(3)] == [True, True, False]


def test_plamo2_matches_written_out_recurrence(tmp_path):
    path = os.path.join(str(tmp_path), "plamo2")
    os.makedirs(path)
    json.dump(CFG, open(os.path.join(path, "config.json"), "w"))
    sd = _weights()
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    prompts = [[5, 17, 99, 3, 250], [(7 * i + 3) % 300 for i in range(37)]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4,
              max_num_batched_tokens=16, block_size=16)
    assert not llm.llm_engine.vllm_config.cache_config.enable_prefix_caching
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=6,
                                                                                   ignore_eos=True))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        ids, ref = list(p), []
        for _ in range(6):
            nxt = int(_dense_forward(sd, ids).argmax())
            ref.append(nxt)
            ids.append(nxt)
        assert list(o.outputs[0].token_ids) == ref, len(p)
