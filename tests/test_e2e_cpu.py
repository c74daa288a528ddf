"""End-to-end CPU plumbing: tiny random HF checkpoints on disk, greedy decode
through LLM.generate must equal transformers' own greedy decode token for token
(the BASELINE "OPT on CPU" correctness configuration, at toy size)."""
import os

import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams


def _save_tiny(tmp, kind):
    import transformers as tf
    torch.manual_seed(0)
    if kind == "llama":
        cfg = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                             num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                             rms_norm_eps=1e-5, tie_word_embeddings=False, head_dim=64)
        m = tf.LlamaForCausalLM(cfg)
    elif kind == "qwen2":
        cfg = tf.Qwen2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                             num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=256,
                             tie_word_embeddings=True)
        m = tf.Qwen2ForCausalLM(cfg)
    elif kind == "opt":
        cfg = tf.OPTConfig(vocab_size=320, hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2,
                           max_position_embeddings=256, word_embed_proj_dim=128)
        m = tf.OPTForCausalLM(cfg)
    elif kind == "gpt2":
        cfg = tf.GPT2Config(vocab_size=320, n_embd=128, n_layer=2, n_head=2, n_positions=256)
        m = tf.GPT2LMHeadModel(cfg)
    elif kind == "mixtral":
        cfg = tf.MixtralConfig(vocab_size=320, hidden_size=128, intermediate_size=128, num_hidden_layers=2,
                               num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                               num_local_experts=4, num_experts_per_tok=2, sliding_window=None, head_dim=64)
        m = tf.MixtralForCausalLM(cfg)
    elif kind == "qwen2_moe":
        cfg = tf.Qwen2MoeConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=256,
                                num_experts=4, num_experts_per_tok=2, moe_intermediate_size=128,
                                shared_expert_intermediate_size=128, decoder_sparse_step=1, norm_topk_prob=False,
                                use_sliding_window=False)
        m = tf.Qwen2MoeForCausalLM(cfg)
    elif kind == "qwen3_moe":
        cfg = tf.Qwen3MoeConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=256,
                                num_experts=4, num_experts_per_tok=2, moe_intermediate_size=128, head_dim=64,
                                decoder_sparse_step=1, norm_topk_prob=True, use_sliding_window=False)
        m = tf.Qwen3MoeForCausalLM(cfg)
    elif kind == "deepseek_v2_mla":
        # latent sizes the GPU MLA kernel is built for (kv_lora_rank 64 + rope 32)
        cfg = tf.DeepseekV2Config(vocab_size=320, hidden_size=128, intermediate_size=256, moe_intermediate_size=128,
                                  num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                                  n_shared_experts=1, n_routed_experts=4, num_experts_per_tok=2,
                                  first_k_dense_replace=1, kv_lora_rank=64, q_lora_rank=48, qk_rope_head_dim=32,
                                  qk_nope_head_dim=32, v_head_dim=32, max_position_embeddings=256,
                                  topk_method="greedy", n_group=1, topk_group=1, norm_topk_prob=False)
        m = tf.DeepseekV2ForCausalLM(cfg)
    elif kind in ("deepseek_v2", "deepseek_v2_yarn"):
        rope = {"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 64, "mscale": 1.0,
                "mscale_all_dim": 1.0, "beta_fast": 32, "beta_slow": 1, "rope_theta": 10000.0} \
            if kind.endswith("yarn") else None
        cfg = tf.DeepseekV2Config(vocab_size=320, hidden_size=128, intermediate_size=256, moe_intermediate_size=64,
                                  num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                                  n_shared_experts=1, n_routed_experts=4, num_experts_per_tok=2,
                                  first_k_dense_replace=1, kv_lora_rank=32, q_lora_rank=48, qk_rope_head_dim=16,
                                  qk_nope_head_dim=32, v_head_dim=32, max_position_embeddings=256,
                                  topk_method="group_limited_greedy", n_group=2, topk_group=1,
                                  routed_scaling_factor=2.0, norm_topk_prob=False, rope_parameters=rope)
        m = tf.DeepseekV2ForCausalLM(cfg)
    elif kind == "deepseek_v3":
        cfg = tf.DeepseekV3Config(vocab_size=320, hidden_size=128, intermediate_size=256, moe_intermediate_size=64,
                                  num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                                  n_shared_experts=1, n_routed_experts=8, num_experts_per_tok=2,
                                  first_k_dense_replace=1, kv_lora_rank=32, q_lora_rank=None, qk_rope_head_dim=16,
                                  qk_nope_head_dim=32, v_head_dim=32, max_position_embeddings=256, n_group=2,
                                  topk_group=1, routed_scaling_factor=2.5, norm_topk_prob=True)
        m = tf.DeepseekV3ForCausalLM(cfg)
        with torch.no_grad():
            for layer in m.model.layers[1:]:
                layer.mlp.gate.e_score_correction_bias.copy_(torch.randn(8) * 0.1)
    elif kind == "gemma":
        cfg = tf.GemmaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                             num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=256)
        m = tf.GemmaForCausalLM(cfg)
    elif kind == "gemma2":
        cfg = tf.Gemma2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                              num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=256,
                              query_pre_attn_scalar=32, sliding_window=16, attn_logit_softcapping=20.0,
                              final_logit_softcapping=10.0)
        m = tf.Gemma2ForCausalLM(cfg)
    elif kind == "phi3":
        cfg = tf.Phi3Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                            num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                            original_max_position_embeddings=256, pad_token_id=0)
        m = tf.Phi3ForCausalLM(cfg)
    elif kind == "granite":
        cfg = tf.GraniteConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                               num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                               embedding_multiplier=6.0, residual_multiplier=0.4, attention_multiplier=0.1,
                               logits_scaling=4.0, tie_word_embeddings=True)
        m = tf.GraniteForCausalLM(cfg)
    elif kind == "gpt_neox":
        cfg = tf.GPTNeoXConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                               num_attention_heads=2, max_position_embeddings=256, rotary_pct=0.25)
        m = tf.GPTNeoXForCausalLM(cfg)
    elif kind == "gptj":
        cfg = tf.GPTJConfig(vocab_size=320, n_embd=128, n_layer=2, n_head=2, n_positions=256, rotary_dim=32)
        m = tf.GPTJForCausalLM(cfg)
    elif kind == "starcoder2":
        cfg = tf.Starcoder2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                  num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                                  sliding_window=None)
        m = tf.Starcoder2ForCausalLM(cfg)
    elif kind == "gpt_bigcode":
        cfg = tf.GPTBigCodeConfig(vocab_size=320, n_embd=128, n_layer=2, n_head=2, n_positions=256, multi_query=True)
        m = tf.GPTBigCodeForCausalLM(cfg)
    elif kind in ("mamba", "falcon_mamba", "mamba2"):
        if kind == "mamba":
            m = tf.MambaForCausalLM(tf.MambaConfig(vocab_size=320, hidden_size=64, state_size=16, num_hidden_layers=2,
                                                   conv_kernel=4, expand=2, use_bias=True))
        elif kind == "falcon_mamba":
            m = tf.FalconMambaForCausalLM(tf.FalconMambaConfig(vocab_size=320, hidden_size=64, state_size=16,
                                                               num_hidden_layers=2, conv_kernel=4, expand=2))
        else:
            m = tf.Mamba2ForCausalLM(tf.Mamba2Config(vocab_size=320, hidden_size=64, state_size=32, num_heads=8,
                                                     head_dim=16, n_groups=2, num_hidden_layers=2, conv_kernel=4,
                                                     expand=2, chunk_size=8, tie_word_embeddings=True))
        with torch.no_grad():      # the default initialisation is so small that greedy decoding repeats one token
            for n, p in m.named_parameters():
                if p.dim() >= 2 and "A_log" not in n:
                    p.mul_(1.5)
    m = m.float().eval()
    path = os.path.join(tmp, kind)
    m.save_pretrained(path, safe_serialization=True)
    return path, m


@pytest.mark.parametrize("kind", ["llama", "opt", "qwen2", "gpt2", "mixtral", "qwen2_moe", "qwen3_moe",
                                  "deepseek_v2", "deepseek_v2_yarn", "deepseek_v2_mla", "deepseek_v3", "gemma", "gemma2", "phi3",
                                  "granite", "gpt_neox", "gptj", "starcoder2", "gpt_bigcode", "mamba",
                                  "falcon_mamba", "mamba2"])
def test_greedy_matches_hf(tmp_path, kind):
    path, hf = _save_tiny(str(tmp_path), kind)
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40, [1, 2, 3]]
    # transformers' eager Mamba-2 normalises the gated output over the whole width, not per group
    extra = {"hf_overrides": {"gated_norm_groups": 1}} if kind == "mamba2" else {}
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False, block_size=16, **extra)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=12,
                                                                                   ignore_eos=True))
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=12, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (kind, p)
    llm.shutdown()


def test_chunked_prefill_prefix_cache_and_preemption_same_tokens(tmp_path):
    path, hf = _save_tiny(str(tmp_path), "llama")
    prompts = [[(i * 7 + j) % 300 for j in range(50)] for i in range(6)] + [[3] * 60, [3] * 60]
    sp = SamplingParams(temperature=0, max_tokens=20, ignore_eos=True)
    base = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
               max_num_seqs=16, max_num_batched_tokens=1024, enable_prefix_caching=False)
    ref = [list(o.outputs[0].token_ids) for o in base.generate(prompt_token_ids=prompts, sampling_params=sp)]
    base.shutdown()
    # small token budget (chunked prefill), prefix caching on, tiny KV pool (forces preemption)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=16, max_num_batched_tokens=48, enable_prefix_caching=True, num_gpu_blocks_override=24)
    got = [list(o.outputs[0].token_ids) for o in llm.generate(prompt_token_ids=prompts, sampling_params=sp)]
    assert got == ref
    llm.shutdown()


def test_deepseek_expanded_heads_path_still_matches_hf(tmp_path, monkeypatch):
    """VDA_MLA=0 keeps the per-head K/V cache (padded heads) instead of the latent cache."""
    monkeypatch.setenv("VDA_MLA", "0")
    path, hf = _save_tiny(str(tmp_path), "deepseek_v2")
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False)
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    assert not any(getattr(m, "is_mla", False) for m in mr.attn_layers.values())
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=8,
                                                                                   ignore_eos=True))
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref
    llm.shutdown()
