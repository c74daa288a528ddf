"""Phi-3-small: no transformers class exists (parity unpinned).  The engine's tokens are compared with the network
written out densely in fp32, with the block-sparse visibility pattern built independently (Kronecker product of
the block mask, as in the reference's vllm/attention/ops/blocksparse_attention/utils.py:163-245), on prompts both
inside and beyond the local span, prefilled in chunks."""
import json
import math
import os

import torch
from safetensors.torch import save_file

from vllm_distributed_amd import LLM, SamplingParams

CFG = dict(architectures=["Phi3SmallForCausalLM"], model_type="phi3small", vocab_size=320, hidden_size=128,
           num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2, ff_intermediate_size=96,
           intermediate_size=96, hidden_act="gegelu", gegelu_limit=2.0, layer_norm_epsilon=1e-5,
           max_position_embeddings=256, rope_embedding_base=10000.0, rope_position_scale=1.0,
           blocksparse_block_size=4, blocksparse_num_local_blocks=2, blocksparse_vert_stride=3,
           blocksparse_homo_head_pattern=False, blocksparse_triton_kernel_block_size=4,
           dense_attention_every_n_layers=2, mup_use_scaling=True, mup_attn_multiplier=1.5,
           mup_embedding_multiplier=4.0, mup_width_multiplier=2.0, tie_word_embeddings=True,
           dummy_token_indices=[310, 311], torch_dtype="float32")


def _weights():
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g) * 0.2                       # noqa: E731
    sd = {"model.embed_tokens.weight": r(320, 128), "model.final_layernorm.weight": 1 + r(128) * 0.1,
          "model.final_layernorm.bias": r(128) * 0.1}
    for i in range(4):
        p = f"model.layers.{i}."
        sd.update({p + "input_layernorm.weight": 1 + r(128) * 0.1, p + "input_layernorm.bias": r(128) * 0.1,
                   p + "post_attention_layernorm.weight": 1 + r(128) * 0.1, p + "post_attention_layernorm.bias": r(128) * 0.1,
                   p + "self_attn.query_key_value.weight": r(2 * 4 * 32, 128), p + "self_attn.query_key_value.bias": r(256) * 0.1,
                   p + "self_attn.dense.weight": r(128, 128), p + "self_attn.dense.bias": r(128) * 0.1,
                   p + "mlp.up_proj.weight": r(192, 128), p + "mlp.up_proj.bias": r(192) * 0.1,
                   p + "mlp.down_proj.weight": r(128, 96), p + "mlp.down_proj.bias": r(128) * 0.1})
    return sd


def _dense_forward(sd, ids):
    """Logits of the last position, every step recomputed from scratch."""
    c = CFG
    T, H, KV, D = len(ids), 4, 2, 32
    x = sd["model.embed_tokens.weight"][ids] * c["mup_embedding_multiplier"]
    pos = torch.arange(T).float()
    inv = 1.0 / 10000.0 ** (torch.arange(0, D, 2).float() / D)
    ang = pos[:, None] * inv[None]
    cos, sin = torch.cat([ang.cos()] * 2, -1), torch.cat([ang.sin()] * 2, -1)

    def rope(t):                                                             # [T, heads, D], rotate-half form
        a, b = t[..., : D // 2], t[..., D // 2:]
        return t * cos[:, None] + torch.cat([-b, a], -1) * sin[:, None]
    bs, local, stride = 4, 2, 3
    nb = math.ceil(T / bs)
    step = max(1, stride // H)
    for i in range(4):
        p = f"model.layers.{i}."
        ln = torch.nn.functional.layer_norm
        h = ln(x, (128,), sd[p + "input_layernorm.weight"], sd[p + "input_layernorm.bias"], 1e-5)
        qkv = (h @ sd[p + "self_attn.query_key_value.weight"].t() + sd[p + "self_attn.query_key_value.bias"])
        qkv = qkv.view(T, KV, 4, D)                                          # per KV group: 2 queries, key, value
        q, k, v = qkv[:, :, :2].reshape(T, H, D), qkv[:, :, 2], qkv[:, :, 3]
        q, k = rope(q), rope(k)
        k, v = k.repeat_interleave(2, 1), v.repeat_interleave(2, 1)
        sc = torch.einsum("thd,shd->hts", q, k) * (c["mup_attn_multiplier"] / D)
        mask = torch.tril(torch.ones(T, T, dtype=torch.bool))[None].expand(H, -1, -1)
        if (i + 1) % 2 != 0:                                                 # block-sparse layer
            qb, kb = torch.arange(nb)[:, None], torch.arange(nb)[None]
            blocks = []
            for hd in range(H):
                stripe = (torch.arange(nb) + hd * step + 1) % stride == 0
                blocks.append((qb >= kb) & ((qb - kb < local) | stripe[None]))
            dense = torch.kron(torch.stack(blocks).float(), torch.ones(bs, bs)).bool()[:, :T, :T]
            mask = mask & dense
        a = torch.softmax(sc.masked_fill(~mask, float("-inf")), -1)
        o = torch.einsum("hts,shd->thd", a, v).reshape(T, 128)
        x = x + o @ sd[p + "self_attn.dense.weight"].t() + sd[p + "self_attn.dense.bias"]
        h = ln(x, (128,), sd[p + "post_attention_layernorm.weight"], sd[p + "post_attention_layernorm.bias"], 1e-5)
        u = h @ sd[p + "mlp.up_proj.weight"].t() + sd[p + "mlp.up_proj.bias"]
        a_, b_ = u[:, ::2].clamp(max=2.0), u[:, 1::2].clamp(-2.0, 2.0)
        x = x + (a_ * torch.sigmoid(1.702 * a_) * (b_ + 1)) @ sd[p + "mlp.down_proj.weight"].t() + sd[p + "mlp.down_proj.bias"]
    x = torch.nn.functional.layer_norm(x, (128,), sd["model.final_layernorm.weight"], sd["model.final_layernorm.bias"], 1e-5)
    logits = x[-1] @ sd["model.embed_tokens.weight"].t() / c["mup_width_multiplier"]
    logits[[310, 311]] = float("-inf")
    return logits


def test_phi3_small_matches_dense_definition(tmp_path):
    path = os.path.join(str(tmp_path), "phi3small")
    os.makedirs(path)
    json.dump(CFG, open(os.path.join(path, "config.json"), "w"))
    sd = _weights()
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    prompts = [[5, 17, 99, 3, 250], [(7 * i + 3) % 300 for i in range(37)]]      # inside / beyond the 8-token span
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4,
              max_num_batched_tokens=16, enable_prefix_caching=False, block_size=16)
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
