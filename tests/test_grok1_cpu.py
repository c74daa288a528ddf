"""Grok-1 (not in transformers: parity with the original code is unpinned): a random checkpoint in its layout --
``attn`` / ``moe_block`` modules, experts ``linear`` / ``linear_v`` / ``linear_1``, norm weights named ``scale`` --
against a plain PyTorch forward written here: soft-capped attention scaled by ``attn_output_multiplier``, GELU-gated
top-2 experts, norms on both sides of each sub-block, embedding and output multipliers."""
import math

import torch

from tests.test_e2e_cpu import _write_checkpoint
from vllm_distributed_amd import LLM, SamplingParams


def _grok_weights(V, d, H, KV, L, F, E):
    torch.manual_seed(3)
    D = d // H
    w = {"model.embed_tokens.weight": torch.randn(V, d) * 0.02, "model.norm.scale": torch.rand(d) + 0.5,
         "lm_head.weight": torch.randn(V, d) * 0.5}
    for i in range(L):
        p = f"model.layers.{i}."
        w[p + "attn.q_proj.weight"] = torch.randn(H * D, d) * 0.3
        w[p + "attn.k_proj.weight"] = torch.randn(KV * D, d) * 0.3
        w[p + "attn.v_proj.weight"] = torch.randn(KV * D, d) * 0.3
        w[p + "attn.o_proj.weight"] = torch.randn(d, H * D) * 1.0
        w[p + "moe_block.gate.weight"] = torch.randn(E, d) * 0.5
        for e in range(E):
            q = p + f"moe_block.experts.{e}."
            w[q + "linear.weight"] = torch.randn(F, d) * 0.2
            w[q + "linear_v.weight"] = torch.randn(F, d) * 0.2
            w[q + "linear_1.weight"] = torch.randn(d, F) * 0.2
        for n in ("pre_attn_norm", "post_attn_norm", "pre_moe_norm", "post_moe_norm"):
            w[p + n + ".scale"] = torch.rand(d) + 0.5
    return w


def test_grok1_matches_plain_torch(tmp_path):
    V, d, H, KV, L, F, E = 320, 128, 4, 2, 2, 256, 4
    D = d // H
    w = _grok_weights(V, d, H, KV, L, F, E)
    cfg = dict(architectures=["Grok1ModelForCausalLM"], model_type="grok-1", vocab_size=V, hidden_size=d,
               intermediate_size=F, num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=KV,
               num_experts=E, num_experts_per_tok=2, rms_norm_eps=1e-5, max_position_embeddings=256,
               rope_theta=10000.0, attn_output_multiplier=0.125, embedding_multiplier_scale=40.0,
               output_multiplier_scale=0.6, attn_logit_softcapping=30.0, tie_word_embeddings=False,
               torch_dtype="float32")
    path = str(tmp_path / "grok1")
    _write_checkpoint(path, w, cfg)

    def rms(x, g):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * g

    def rope(x, pos):                       # x [T, heads, D], rotate-half form
        inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
        f = pos[:, None].float() * inv[None]
        cos, sin = torch.cat([f.cos(), f.cos()], -1)[:, None], torch.cat([f.sin(), f.sin()], -1)[:, None]
        x1, x2 = x[..., : D // 2], x[..., D // 2:]
        return x * cos + torch.cat([-x2, x1], -1) * sin

    def forward(ids):
        T = len(ids)
        pos = torch.arange(T)
        x = w["model.embed_tokens.weight"][ids] * 40.0
        mask = torch.full((T, T), float("-inf")).triu(1)
        for i in range(L):
            p = f"model.layers.{i}."
            h = rms(x, w[p + "pre_attn_norm.scale"])
            q = rope((h @ w[p + "attn.q_proj.weight"].T).view(T, H, D), pos).transpose(0, 1)
            k = rope((h @ w[p + "attn.k_proj.weight"].T).view(T, KV, D), pos).transpose(0, 1)
            v = (h @ w[p + "attn.v_proj.weight"].T).view(T, KV, D).transpose(0, 1)
            k, v = k.repeat_interleave(H // KV, 0), v.repeat_interleave(H // KV, 0)
            s = q @ k.transpose(1, 2) * D ** -0.5
            s = 30.0 * torch.tanh(s / 30.0) + mask
            a = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(T, H * D) @ w[p + "attn.o_proj.weight"].T
            x = x + rms(a * 0.125, w[p + "post_attn_norm.scale"])
            h = rms(x, w[p + "pre_moe_norm.scale"])
            probs = torch.softmax(h @ w[p + "moe_block.gate.weight"].T, -1)
            tw, ti = probs.topk(2, -1)
            tw = tw / tw.sum(-1, keepdim=True)
            out = torch.zeros_like(h)
            for t in range(T):
                for j in range(2):
                    e = int(ti[t, j])
                    q_ = p + f"moe_block.experts.{e}."
                    g = torch.nn.functional.gelu(h[t] @ w[q_ + "linear.weight"].T) * (h[t] @ w[q_ + "linear_v.weight"].T)
                    out[t] += tw[t, j] * (g @ w[q_ + "linear_1.weight"].T)
            x = x + rms(out, w[p + "post_moe_norm.scale"])
        return (rms(x, w["model.norm.scale"]) @ w["lm_head.weight"].T) * 0.6

    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 10]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False)
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, logprobs=0))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        gen = list(o.outputs[0].token_ids)
        lp = torch.log_softmax(forward(torch.tensor(p + gen)), -1)
        worst = max(abs(float(lp[len(p) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob) for k, t in enumerate(gen))
        assert worst < 2e-3, worst
        assert gen[0] == int(lp[len(p) - 1].argmax())
