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
    elif kind in ("llama4", "llama4_top2"):
        # rotary + chunked layers (chunks of 32 positions) and a global layer without positions; layers 1 and 3 are MoE
        cfg = tf.Llama4TextConfig(vocab_size=320, hidden_size=128, intermediate_size=128, intermediate_size_mlp=256,
                                  num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1, head_dim=64,
                                  max_position_embeddings=256, num_local_experts=4, interleave_moe_layer_step=2,
                                  num_experts_per_tok=2 if kind == "llama4_top2" else 1,
                                  attention_chunk_size=32, floor_scale=16, attn_scale=0.1, pad_token_id=0,
                                  bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
        m = tf.Llama4ForCausalLM(cfg)
        with torch.no_grad():       # the default initialisation leaves the router indifferent
            for n, p_ in m.named_parameters():
                if "router" in n or "experts" in n:
                    p_.normal_(0, 0.08)
    elif kind == "ernie45":
        cfg = tf.Ernie4_5Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=256,
                                use_bias=True, tie_word_embeddings=True)
        m = tf.Ernie4_5ForCausalLM(cfg)
    elif kind == "ernie45_moe":
        cfg = tf.Ernie4_5_MoeConfig(vocab_size=320, hidden_size=128, intermediate_size=256, moe_intermediate_size=64,
                                    num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=1,
                                    max_position_embeddings=256, moe_num_experts=4, moe_k=2, moe_num_shared_experts=1,
                                    moe_layer_start_index=1, moe_layer_interval=1, tie_word_embeddings=True)
        m = tf.Ernie4_5_MoeForCausalLM(cfg)
        with torch.no_grad():
            for n, p_ in m.named_parameters():
                if "e_score_correction_bias" in n:
                    p_.normal_(0, 0.2)
    elif kind == "dots1":
        cfg = tf.Dots1Config(vocab_size=320, hidden_size=128, intermediate_size=256, moe_intermediate_size=64,
                             num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=1, head_dim=64,
                             n_routed_experts=8, n_shared_experts=1, num_experts_per_tok=2, first_k_dense_replace=1,
                             n_group=2, topk_group=1, norm_topk_prob=True, routed_scaling_factor=1.5,
                             max_position_embeddings=256)
        m = tf.Dots1ForCausalLM(cfg)
        with torch.no_grad():
            for n, b in m.named_buffers():
                if "e_score_correction_bias" in n:
                    b.normal_(0, 0.2)
    elif kind in ("hunyuan_dense", "hunyuan_moe"):
        kw = dict(vocab_size=320, hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                  num_key_value_heads=1, head_dim=64, max_position_embeddings=256)
        if kind == "hunyuan_moe":
            m = tf.HunYuanMoEV1ForCausalLM(tf.HunYuanMoEV1Config(num_experts=4, moe_topk=2, **kw))
        else:
            m = tf.HunYuanDenseV1ForCausalLM(tf.HunYuanDenseV1Config(**kw))
    elif kind == "persimmon":
        cfg = tf.PersimmonConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                 num_attention_heads=2, max_position_embeddings=256)
        m = tf.PersimmonForCausalLM(cfg)
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
    elif kind == "phimoe":
        cfg = tf.PhimoeConfig(vocab_size=320, hidden_size=128, intermediate_size=128, num_hidden_layers=2,
                              num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                              num_local_experts=4, num_experts_per_tok=2, sliding_window=None, attention_bias=True,
                              lm_head_bias=True, router_jitter_noise=0.01, input_jitter_noise=0.0)
        m = tf.PhimoeForCausalLM(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2:
                    p.mul_(3.0)
    elif kind == "dbrx":
        cfg = tf.DbrxConfig(d_model=128, n_heads=2, n_layers=2, max_seq_len=256, vocab_size=320,
                            attn_config=dict(kv_n_heads=1, clip_qkv=1.5, rope_theta=10000.0),
                            ffn_config=dict(ffn_hidden_size=128, moe_num_experts=4, moe_top_k=2,
                                            moe_normalize_expert_weights=1.0))
        m = tf.DbrxForCausalLM(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2:
                    p.mul_(3.0)
    elif kind == "olmoe":
        cfg = tf.OlmoeConfig(vocab_size=320, hidden_size=128, intermediate_size=64, num_hidden_layers=2,
                             num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                             num_experts=4, num_experts_per_tok=2)
        m = tf.OlmoeForCausalLM(cfg)
    elif kind == "granitemoe":
        cfg = tf.GraniteMoeConfig(vocab_size=320, hidden_size=128, intermediate_size=64, num_hidden_layers=2,
                                  num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                                  num_local_experts=4, num_experts_per_tok=2, embedding_multiplier=6.0,
                                  residual_multiplier=0.4, attention_multiplier=0.1, logits_scaling=4.0)
        m = tf.GraniteMoeForCausalLM(cfg)
    elif kind == "jamba":
        cfg = tf.JambaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=4,
                             num_attention_heads=2, num_key_value_heads=1, attn_layer_period=2, attn_layer_offset=1,
                             expert_layer_period=2, expert_layer_offset=1, num_experts=4, num_experts_per_tok=2,
                             mamba_d_state=16, mamba_d_conv=4, mamba_expand=2, mamba_dt_rank=8,
                             use_mamba_kernels=False, max_position_embeddings=256)
        m = tf.JambaForCausalLM(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2 and "A_log" not in n:
                    p.mul_(1.5)
    elif kind == "minimax":
        cfg = tf.MiniMaxConfig(vocab_size=320, hidden_size=128, intermediate_size=192, num_hidden_layers=4,
                               num_attention_heads=2, num_key_value_heads=1, head_dim=64, num_local_experts=4,
                               num_experts_per_tok=2, max_position_embeddings=512, block_size=16,
                               layer_types=["linear_attention", "full_attention", "linear_attention", "linear_attention"],
                               full_attn_alpha_factor=1.5, full_attn_beta_factor=0.8, linear_attn_alpha_factor=1.3,
                               linear_attn_beta_factor=0.9, mlp_alpha_factor=1.2, mlp_beta_factor=0.7)
        m = tf.MiniMaxForCausalLM(cfg)
    elif kind == "falcon_h1":
        cfg = tf.FalconH1Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                                num_attention_heads=2, num_key_value_heads=1, head_dim=64, mamba_d_ssm=128,
                                mamba_n_heads=4, mamba_d_head=32, mamba_n_groups=1, mamba_d_state=16, mamba_d_conv=4,
                                mamba_expand=1, mamba_chunk_size=16, mamba_rms_norm=False, max_position_embeddings=256,
                                embedding_multiplier=1.7, lm_head_multiplier=0.6, mlp_multipliers=[1.3, 0.7],
                                key_multiplier=0.8, attention_in_multiplier=1.2, attention_out_multiplier=0.9,
                                ssm_multipliers=[0.9, 1.1, 1.2, 0.8, 1.3], ssm_in_multiplier=1.1,
                                ssm_out_multiplier=0.85)
        m = tf.FalconH1ForCausalLM(cfg)
    elif kind == "zamba2":
        # hybrid layer ids chosen so that transformers' block id (layer index mod num_mem_blocks) agrees with the
        # order in which the shared blocks are taken in turn
        cfg = tf.Zamba2Config(vocab_size=320, hidden_size=64, num_hidden_layers=5,
                              layers_block_type=["hybrid", "mamba", "mamba", "hybrid", "hybrid"], mamba_d_state=16,
                              mamba_d_conv=4, mamba_expand=2, mamba_ngroups=2, n_mamba_heads=4, num_attention_heads=2,
                              num_key_value_heads=2, intermediate_size=128, num_mem_blocks=2, adapter_rank=8,
                              use_shared_attention_adapter=True, use_mem_rope=True, max_position_embeddings=256,
                              chunk_size=8, pad_token_id=0, use_mamba_kernels=False)
        m = tf.Zamba2ForCausalLM(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2 and "A_log" not in n:
                    p.mul_(2.0)
    elif kind == "gemma3n":
        cfg = tf.Gemma3nTextConfig(vocab_size=320, vocab_size_per_layer_input=320, hidden_size=128,
                                   hidden_size_per_layer_input=16, intermediate_size=[192] * 6, num_hidden_layers=6,
                                   num_attention_heads=2, num_key_value_heads=1, head_dim=64,
                                   max_position_embeddings=256, sliding_window=16,
                                   layer_types=["sliding_attention", "sliding_attention", "full_attention"] * 2,
                                   altup_num_inputs=3, altup_active_idx=0, laurel_rank=8, num_kv_shared_layers=2,
                                   activation_sparsity_pattern=[0.9, 0.9, 0.0, 0.0, 0.0, 0.0], pad_token_id=0,
                                   final_logit_softcapping=30.0)
        m = tf.Gemma3nForCausalLM(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2:
                    p.mul_(3.0)
                elif "correct_output_scale" in n:
                    p.normal_(0, 0.5)
    elif kind == "bamba":
        cfg = tf.BambaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=4,
                             num_attention_heads=2, num_key_value_heads=1, attn_layer_indices=[1, 3], mamba_n_heads=8,
                             mamba_d_head=32, mamba_n_groups=2, mamba_d_state=16, mamba_d_conv=4, mamba_expand=2,
                             mamba_chunk_size=8, max_position_embeddings=256)
        m = tf.BambaForCausalLM(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2 and "A_log" not in n:
                    p.mul_(1.5)
    elif kind == "granitemoehybrid":
        cfg = tf.GraniteMoeHybridConfig(
            vocab_size=320, hidden_size=128, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
            num_key_value_heads=1, max_position_embeddings=256, num_local_experts=4, num_experts_per_tok=2,
            shared_intermediate_size=96, layer_types=["mamba", "attention", "mamba"], mamba_n_heads=8, mamba_d_head=32,
            mamba_n_groups=2, mamba_d_state=16, mamba_d_conv=4, mamba_expand=2, mamba_chunk_size=8,
            embedding_multiplier=6.0, residual_multiplier=0.4, attention_multiplier=0.1, logits_scaling=4.0,
            position_embedding_type="rope")
        m = tf.GraniteMoeHybridForCausalLM(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2 and "A_log" not in n:
                    p.mul_(1.5)
    elif kind == "nemotron_h":
        cfg = tf.NemotronHConfig(vocab_size=320, hidden_size=128, layers_block_type=["mamba", "mlp", "attention", "mlp"],
                                 num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=256,
                                 intermediate_size=256, ssm_state_size=16, mamba_num_heads=8, mamba_head_dim=32,
                                 n_groups=2, conv_kernel=4, chunk_size=8, use_mamba_kernels=False)
        m = tf.NemotronHForCausalLM(cfg)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2 and "A_log" not in n:
                    p.mul_(1.5)
    elif kind == "stablelm":
        cfg = tf.StableLmConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                                use_qkv_bias=True)
        m = tf.StableLmForCausalLM(cfg)
    elif kind == "olmo":
        cfg = tf.OlmoConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                            num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, clip_qkv=0.5)
        m = tf.OlmoForCausalLM(cfg)
    elif kind == "olmo2":
        cfg = tf.Olmo2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                             num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256)
        m = tf.Olmo2ForCausalLM(cfg)
    elif kind in ("cohere", "cohere_qknorm"):
        cfg = tf.CohereConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                              num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                              logit_scale=0.25, use_qk_norm=kind == "cohere_qknorm")
        m = tf.CohereForCausalLM(cfg)
        if kind == "cohere_qknorm":
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if "q_norm" in n or "k_norm" in n:
                        p.uniform_(0.5, 1.5)
    elif kind == "cohere2":
        cfg = tf.Cohere2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=4,
                               num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                               logit_scale=0.25, sliding_window=16,
                               layer_types=["sliding_attention", "sliding_attention", "sliding_attention",
                                            "full_attention"])
        m = tf.Cohere2ForCausalLM(cfg)
    elif kind == "granitemoeshared":
        cfg = tf.GraniteMoeSharedConfig(vocab_size=320, hidden_size=128, intermediate_size=64, num_hidden_layers=2,
                                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                                        num_local_experts=4, num_experts_per_tok=2, shared_intermediate_size=96,
                                        embedding_multiplier=6.0, residual_multiplier=0.4, attention_multiplier=0.1,
                                        logits_scaling=4.0)
        m = tf.GraniteMoeSharedForCausalLM(cfg)
    elif kind == "nemotron":
        cfg = tf.NemotronConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                                head_dim=64)
        m = tf.NemotronForCausalLM(cfg)
    elif kind == "glm":
        cfg = tf.GlmConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                           num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, head_dim=64,
                           pad_token_id=0)
        m = tf.GlmForCausalLM(cfg)
    elif kind == "glm4":
        cfg = tf.Glm4Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                            num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, head_dim=64,
                            pad_token_id=0)
        m = tf.Glm4ForCausalLM(cfg)
    elif kind == "gemma3":
        cfg = tf.Gemma3TextConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                                  num_attention_heads=2, num_key_value_heads=1, head_dim=64,
                                  max_position_embeddings=256, query_pre_attn_scalar=32, sliding_window=16,
                                  layer_types=["sliding_attention", "sliding_attention", "full_attention"])
        m = tf.Gemma3ForCausalLM(cfg)
    elif kind in ("falcon_new", "falcon_mq", "falcon_alibi"):
        kw = {"falcon_new": dict(new_decoder_architecture=True, num_kv_heads=1),
              "falcon_mq": dict(new_decoder_architecture=False, multi_query=True, parallel_attn=True),
              "falcon_alibi": dict(new_decoder_architecture=False, multi_query=False, parallel_attn=False,
                                   alibi=True, bias=True)}[kind]
        cfg = tf.FalconConfig(vocab_size=320, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                              max_position_embeddings=256, **kw)
        m = tf.FalconForCausalLM(cfg)
    elif kind == "bloom":
        cfg = tf.BloomConfig(vocab_size=320, hidden_size=128, n_layer=2, n_head=2)
        m = tf.BloomForCausalLM(cfg)
    elif kind == "mpt":
        cfg = tf.MptConfig(vocab_size=320, d_model=128, n_layers=2, n_heads=2, max_seq_len=256)
        m = tf.MptForCausalLM(cfg)
    elif kind == "phi":
        cfg = tf.PhiConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                           num_attention_heads=2, max_position_embeddings=256)
        m = tf.PhiForCausalLM(cfg)
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
                                  "falcon_mamba", "mamba2", "stablelm", "olmo", "olmo2", "cohere", "cohere_qknorm", "nemotron",
                                  "glm", "glm4", "gemma3", "falcon_new", "falcon_mq", "falcon_alibi", "bloom",
                                  "mpt", "phi", "olmoe", "granitemoe", "jamba", "minimax", "falcon_h1", "cohere2", "granitemoeshared", "bamba", "zamba2", "gemma3n", "phimoe", "dbrx", "granitemoehybrid", "nemotron_h", "llama4", "llama4_top2", "ernie45", "ernie45_moe",
                                  "dots1", "hunyuan_dense", "hunyuan_moe", "persimmon"])
def test_greedy_matches_hf(tmp_path, kind):
    path, hf = _save_tiny(str(tmp_path), kind)
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40, [1, 2, 3]]
    # transformers' eager Mamba-2 normalises the gated output over the whole width, not per group
    extra = {"hf_overrides": {"gated_norm_groups": 1}} if kind in ("mamba2", "bamba", "granitemoehybrid") else {}
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


@pytest.mark.parametrize("head_dim", [32, 80, 96, 112, 192])
def test_head_sizes_outside_the_kernel_set_match_hf(tmp_path, head_dim):
    """Heads of 80 / 96 / 112 / 192 (and small ones) run zero-padded to 64 / 128 / 256."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.LlamaConfig(vocab_size=320, hidden_size=2 * head_dim, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, head_dim=head_dim,
                         tie_word_embeddings=False)
    hf = tf.LlamaForCausalLM(cfg).float().eval()
    path = os.path.join(tmp_path, "m")
    hf.save_pretrained(path, safe_serialization=True)
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64, enable_prefix_caching=False, block_size=16)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=10,
                                                                                   ignore_eos=True))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=10, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref


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


def test_llama4_chunked_attention_with_chunked_prefill(tmp_path):
    """Prefill pieces that start and end in the middle of attention chunks, decode steps that cross chunk
    boundaries, prefix-cache hits that skip whole chunks: same tokens as transformers."""
    path, hf = _save_tiny(str(tmp_path), "llama4")
    prompts = [[(i * 13 + j * 7) % 300 + 3 for j in range(n)] for i, n in enumerate((70, 33, 31, 100, 64))]
    prompts.append(prompts[0][:48] + [9, 8, 7])
    sp = SamplingParams(temperature=0, max_tokens=40, ignore_eos=True)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=24, enable_prefix_caching=True, block_size=16)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=sp)
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=40, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, len(p)
    llm.shutdown()


def test_deepseek_fp8_latent_cache(tmp_path):
    """kv_cache_dtype=fp8 with MLA: the latent cache holds e4m3 rows and generation follows the 16-bit-cache run
    (teacher-forced log-probabilities within fp8 tolerance)."""
    path, hf = _save_tiny(str(tmp_path), "deepseek_v2")
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False, kv_cache_dtype="fp8")
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    assert all(getattr(m, "is_mla", False) and m.k_cache.dtype == torch.float8_e4m3fn for m in mr.attn_layers.values())
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, logprobs=0))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        gen = list(o.outputs[0].token_ids)
        with torch.no_grad():
            lp = torch.log_softmax(hf(torch.tensor([p + gen])).logits[0].float(), -1)
        worst = max(abs(float(lp[len(p) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob) for k, t in enumerate(gen))
        assert worst < 0.2, worst


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


def _write_checkpoint(path, state, config: dict):
    import json
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    save_file({k: v.contiguous() for k, v in state.items()}, os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(config, f)


def test_internlm2_layout_equals_llama(tmp_path):
    """InternLM2 is not in transformers: the same weights as a tiny HF Llama, renamed and with q/k/v
    interleaved per KV head the way InternLM2 checkpoints store them, must decode the same tokens."""
    _, hf = _save_tiny(str(tmp_path), "llama")
    c = hf.config
    H, KV, D = c.num_attention_heads, c.num_key_value_heads, c.head_dim
    g = H // KV
    names = {"self_attn.o_proj": "attention.wo", "input_layernorm": "attention_norm",
             "post_attention_layernorm": "ffn_norm", "mlp.gate_proj": "feed_forward.w1",
             "mlp.up_proj": "feed_forward.w3", "mlp.down_proj": "feed_forward.w2"}
    state, sd = {}, hf.state_dict()
    for k, v in sd.items():
        if ".q_proj." in k:
            q, kk, vv = v, sd[k.replace("q_proj", "k_proj")], sd[k.replace("q_proj", "v_proj")]
            fused = torch.cat([q.view(KV, g, D, -1), kk.view(KV, 1, D, -1), vv.view(KV, 1, D, -1)], dim=1)
            state[k.replace("self_attn.q_proj", "attention.wqkv")] = fused.reshape((H + 2 * KV) * D, -1)
            continue
        if ".k_proj." in k or ".v_proj." in k:
            continue
        k = k.replace("model.embed_tokens", "model.tok_embeddings").replace("lm_head.", "output.")
        for a, b in names.items():
            k = k.replace(a, b)
        state[k] = v
    cfg = c.to_dict()
    cfg.update(architectures=["InternLM2ForCausalLM"], model_type="internlm2", rope_theta=10000.0)
    path = str(tmp_path / "internlm2")
    _write_checkpoint(path, state, cfg)
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=10,
                                                                                   ignore_eos=True))
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=10, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref
    llm.shutdown()


def test_baichuan_alibi_matches_plain_torch(tmp_path):
    """Baichuan is not in transformers either: a random checkpoint in its layout (W_pack, ALiBi, no
    rotary) against a plain PyTorch forward written here.  Parity with the original code is unpinned."""
    from vllm_distributed_amd.models.llama_variants import alibi_slopes
    torch.manual_seed(1)
    V, d, H, L, F = 320, 128, 2, 2, 256
    D = d // H
    w = {"model.embed_tokens.weight": torch.randn(V, d) * 0.5, "model.norm.weight": torch.rand(d) + 0.5,
         "lm_head.weight": torch.randn(V, d) * 0.2}
    for i in range(L):
        p = f"model.layers.{i}."
        w[p + "self_attn.W_pack.weight"] = torch.randn(3 * d, d) * 0.15
        w[p + "self_attn.o_proj.weight"] = torch.randn(d, d) * 0.1
        w[p + "mlp.gate_proj.weight"] = torch.randn(F, d) * 0.1
        w[p + "mlp.up_proj.weight"] = torch.randn(F, d) * 0.1
        w[p + "mlp.down_proj.weight"] = torch.randn(d, F) * 0.1
        w[p + "input_layernorm.weight"] = torch.rand(d) + 0.5
        w[p + "post_attention_layernorm.weight"] = torch.rand(d) + 0.5
    cfg = dict(architectures=["BaiChuanForCausalLM"], model_type="baichuan", vocab_size=V, hidden_size=d,
               intermediate_size=F, num_hidden_layers=L, num_attention_heads=H, rms_norm_eps=1e-6,
               model_max_length=256, max_position_embeddings=256, tie_word_embeddings=False, torch_dtype="float32")
    path = str(tmp_path / "baichuan")
    _write_checkpoint(path, w, cfg)

    def rms(x, g):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * g

    slopes = torch.tensor(alibi_slopes(H))

    def forward(ids):
        x = w["model.embed_tokens.weight"][ids]
        T = len(ids)
        pos = torch.arange(T)
        bias = slopes[:, None, None] * (pos[None, :] - pos[:, None])[None].float()
        mask = torch.full((T, T), float("-inf")).triu(1)
        for i in range(L):
            p = f"model.layers.{i}."
            q, k, v = (rms(x, w[p + "input_layernorm.weight"]) @ w[p + "self_attn.W_pack.weight"].T).chunk(3, -1)
            q, k, v = (t.view(T, H, D).transpose(0, 1) for t in (q, k, v))
            a = torch.softmax(q @ k.transpose(1, 2) * D ** -0.5 + bias + mask, -1) @ v
            x = x + a.transpose(0, 1).reshape(T, d) @ w[p + "self_attn.o_proj.weight"].T
            h = rms(x, w[p + "post_attention_layernorm.weight"])
            x = x + (torch.nn.functional.silu(h @ w[p + "mlp.gate_proj.weight"].T) *
                     (h @ w[p + "mlp.up_proj.weight"].T)) @ w[p + "mlp.down_proj.weight"].T
        return rms(x, w["model.norm.weight"])[-1] @ w["lm_head.weight"].T

    prompts = [[5, 17, 99, 3, 250, 11], [7] * 30]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=8,
                                                                                   ignore_eos=True))
    for p, o in zip(prompts, outs):
        ids = list(p)
        for _ in range(8):
            ids.append(int(forward(torch.tensor(ids)).argmax()))
        assert list(o.outputs[0].token_ids) == ids[len(p):]
    llm.shutdown()


def test_cascade_attention_same_tokens(tmp_path, monkeypatch):
    """Requests that share a cached prefix take the shared-prefix + suffix + merge path; tokens must
    equal those of the ordinary path."""
    from vllm_distributed_amd.layers.attention import Attention
    monkeypatch.setenv("VDA_CASCADE_MIN_PREFIX", "32")
    monkeypatch.setenv("VDA_CASCADE_MIN_SEQS", "2")
    path, _ = _save_tiny(str(tmp_path), "llama")
    prefix = [(i * 13) % 300 + 1 for i in range(70)]
    prompts = [prefix + [5 + i, 9, 11 + 2 * i] * (i + 1) for i in range(4)]
    sp = SamplingParams(temperature=0, max_tokens=10, ignore_eos=True)
    calls = []
    orig = Attention._cascade
    monkeypatch.setattr(Attention, "_cascade", lambda self, *a, **k: (calls.append(a[2].cascade_prefix_len),
                                                                      orig(self, *a, **k))[1])
    res = {}
    for name, off in (("cascade", False), ("plain", True)):
        llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
                  max_num_seqs=8, max_num_batched_tokens=128, enable_prefix_caching=True, block_size=16,
                  disable_cascade_attn=off)
        llm.generate(prompt_token_ids=[prefix + [1, 2]], sampling_params=SamplingParams(temperature=0, max_tokens=1))
        n0 = len(calls)
        res[name] = [list(o.outputs[0].token_ids) for o in llm.generate(prompt_token_ids=prompts, sampling_params=sp)]
        res[name + "_calls"] = len(calls) - n0
        llm.shutdown()
    assert res["cascade_calls"] > 0 and res["plain_calls"] == 0
    assert all(p == 64 for p in calls)              # four whole blocks of the 70-token prefix
    assert res["cascade"] == res["plain"]


def test_gemma3_multimodal_checkpoint_served_as_text(tmp_path):
    """A Gemma3ForConditionalGeneration checkpoint: the decoder is loaded from under ``language_model``, the
    vision tower is ignored, and text generation equals transformers' on the same checkpoint."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.Gemma3Config(
        text_config=dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                         num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=256,
                         query_pre_attn_scalar=32, sliding_window=16,
                         layer_types=["sliding_attention", "sliding_attention", "full_attention"]),
        vision_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                           image_size=28, patch_size=14), mm_tokens_per_image=4, image_token_index=300)
    hf = tf.Gemma3ForConditionalGeneration(cfg).float().eval()
    path = os.path.join(str(tmp_path), "gemma3mm")
    hf.save_pretrained(path, safe_serialization=True)
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, enable_prefix_caching=False)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=10,
                                                                                   ignore_eos=True))
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([p]), max_new_tokens=10, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref
    llm.shutdown()


def test_fairseq2_llama_checkpoint(tmp_path):
    """A Llama checkpoint re-written the way fairseq2 stores it (its names, the interleaved rotary row order of the
    q / k projections, the {"model_key": ...} wrapper in a .pt file) decodes the same tokens as the original."""
    import json
    from safetensors.torch import load_file
    path, hf = _save_tiny(str(tmp_path), "llama")
    sd = load_file(os.path.join(path, "model.safetensors"))
    hd = 64
    ren = (("model.embed_tokens.", "decoder_frontend.embed."), ("lm_head.", "final_proj."), ("model.", "decoder."))
    sub = ((".input_layernorm.", ".self_attn_layer_norm."), (".post_attention_layernorm.", ".ffn_layer_norm."),
           (".self_attn.o_proj.", ".self_attn.output_proj."), (".mlp.gate_proj.", ".ffn.gate_proj."),
           (".mlp.up_proj.", ".ffn.inner_proj."), (".mlp.down_proj.", ".ffn.output_proj."), ("decoder.norm.", "decoder.layer_norm."))
    out = {}
    for n, t in sd.items():
        if ".q_proj." in n or ".k_proj." in n:          # half-split -> interleaved pairs
            heads = t.shape[0] // hd
            t = t.view(heads, 2, hd // 2, t.shape[1]).transpose(1, 2).reshape(t.shape).contiguous()
        for a, b in ren:
            if n.startswith(a):
                n = b + n[len(a):]
                break
        for a, b in sub:
            n = n.replace(a, b)
        out[n] = t
    fs2 = os.path.join(str(tmp_path), "fs2")
    os.makedirs(fs2)
    torch.save({"model_key": "model", "model": out}, os.path.join(fs2, "model.pt"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["architectures"] = ["Fairseq2LlamaForCausalLM"]
    json.dump(cfg, open(os.path.join(fs2, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40]
    llm = LLM(model=fs2, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=10,
                                                                                   ignore_eos=True))
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=10, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref
