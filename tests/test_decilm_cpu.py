"""DeciLM / Nemotron-NAS (remote code upstream, not in transformers: parity with the original code is unpinned): a
random checkpoint whose ``block_configs`` give every layer its own shape -- different KV-head counts, a layer without
attention, a layer without MLP, different MLP widths -- against a plain PyTorch forward written here."""
import torch

from tests.test_e2e_cpu import _write_checkpoint
from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.models.llama_variants import deci_intermediate_size


def test_decilm_matches_plain_torch(tmp_path):
    torch.manual_seed(5)
    V, d, H, D = 320, 256, 4, 64
    blocks = [{"attention": {"n_heads_in_group": 2, "no_op": False}, "ffn": {"ffn_mult": 1.0, "no_op": False}},
              {"attention": {"n_heads_in_group": None, "no_op": True}, "ffn": {"ffn_mult": 2.5, "no_op": False}},
              {"attention": {"n_heads_in_group": 4, "no_op": False}, "ffn": {"ffn_mult": None, "no_op": True}},
              {"attention": {"n_heads_in_group": 1, "no_op": False}, "ffn": {"ffn_mult": 0.5, "no_op": False}}]
    w = {"model.embed_tokens.weight": torch.randn(V, d) * 0.5, "model.norm.weight": torch.rand(d) + 0.5,
         "lm_head.weight": torch.randn(V, d) * 0.2}
    shapes = []
    for i, b in enumerate(blocks):
        p = f"model.layers.{i}."
        kv = None
        if not b["attention"]["no_op"]:
            kv = H // b["attention"]["n_heads_in_group"]
            w[p + "self_attn.q_proj.weight"] = torch.randn(H * D, d) * 0.1
            w[p + "self_attn.k_proj.weight"] = torch.randn(kv * D, d) * 0.1
            w[p + "self_attn.v_proj.weight"] = torch.randn(kv * D, d) * 0.1
            w[p + "self_attn.o_proj.weight"] = torch.randn(d, H * D) * 0.1
            w[p + "input_layernorm.weight"] = torch.rand(d) + 0.5
        F = None
        if not b["ffn"]["no_op"]:
            F = deci_intermediate_size(b["ffn"]["ffn_mult"], d)
            w[p + "mlp.gate_proj.weight"] = torch.randn(F, d) * 0.1
            w[p + "mlp.up_proj.weight"] = torch.randn(F, d) * 0.1
            w[p + "mlp.down_proj.weight"] = torch.randn(d, F) * 0.05
            w[p + "post_attention_layernorm.weight"] = torch.rand(d) + 0.5
        shapes.append((kv, F))
    assert [s[1] for s in shapes] == [256, 512, None, 256] and [s[0] for s in shapes] == [2, None, 1, 4]
    cfg = dict(architectures=["DeciLMForCausalLM"], model_type="nemotron-nas", vocab_size=V, hidden_size=d,
               num_hidden_layers=len(blocks), num_attention_heads=H, block_configs=blocks, rms_norm_eps=1e-5,
               max_position_embeddings=256, rope_theta=10000.0, hidden_act="silu", tie_word_embeddings=False,
               torch_dtype="float32", intermediate_size=256)
    path = str(tmp_path / "deci")
    _write_checkpoint(path, w, cfg)

    def rms(x, g):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * g

    def rope(x, pos):
        inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
        f = pos[:, None].float() * inv[None]
        cos, sin = torch.cat([f.cos(), f.cos()], -1)[:, None], torch.cat([f.sin(), f.sin()], -1)[:, None]
        return x * cos + torch.cat([-x[..., D // 2:], x[..., : D // 2]], -1) * sin

    def forward(ids):
        T = len(ids)
        pos = torch.arange(T)
        x = w["model.embed_tokens.weight"][ids]
        mask = torch.full((T, T), float("-inf")).triu(1)
        for i, (kv, F) in enumerate(shapes):
            p = f"model.layers.{i}."
            if kv is not None:
                h = rms(x, w[p + "input_layernorm.weight"])
                q = rope((h @ w[p + "self_attn.q_proj.weight"].T).view(T, H, D), pos).transpose(0, 1)
                k = rope((h @ w[p + "self_attn.k_proj.weight"].T).view(T, kv, D), pos).transpose(0, 1)
                v = (h @ w[p + "self_attn.v_proj.weight"].T).view(T, kv, D).transpose(0, 1)
                k, v = k.repeat_interleave(H // kv, 0), v.repeat_interleave(H // kv, 0)
                a = torch.softmax(q @ k.transpose(1, 2) * D ** -0.5 + mask, -1) @ v
                x = x + a.transpose(0, 1).reshape(T, H * D) @ w[p + "self_attn.o_proj.weight"].T
            if F is not None:
                h = rms(x, w[p + "post_attention_layernorm.weight"])
                x = x + (torch.nn.functional.silu(h @ w[p + "mlp.gate_proj.weight"].T)
                         * (h @ w[p + "mlp.up_proj.weight"].T)) @ w[p + "mlp.down_proj.weight"].T
        return rms(x, w["model.norm.weight"]) @ w["lm_head.weight"].T

    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 12]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False)
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    assert sorted(m.k_cache.shape[1] for m in mr.attn_layers.values()) == [1, 2, 4]      # KV heads per layer
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, logprobs=0))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        gen = list(o.outputs[0].token_ids)
        lp = torch.log_softmax(forward(torch.tensor(p + gen)), -1)
        worst = max(abs(float(lp[len(p) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob) for k, t in enumerate(gen))
        assert worst < 2e-3, worst
        assert gen[0] == int(lp[len(p) - 1].argmax())
