"""Architecture presets for well-known checkpoints.

The target machines have no hub access, so ``LLM("meta-llama/Meta-Llama-3-70B",
load_format="dummy")`` must work from a name alone.  Values are the public
``config.json`` contents of each model family.
"""

def _llama(hidden, inter, layers, heads, kv, vocab=128256, theta=500000.0, maxpos=8192,
           rope_scaling=None, tie=False, arch="LlamaForCausalLM", **kw):
    d = dict(architectures=[arch], model_type="llama", hidden_size=hidden, intermediate_size=inter,
             num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=kv,
             vocab_size=vocab, rms_norm_eps=1e-5, rope_theta=theta, max_position_embeddings=maxpos,
             hidden_act="silu", tie_word_embeddings=tie, torch_dtype="bfloat16",
             bos_token_id=128000, eos_token_id=128001, attention_bias=False, mlp_bias=False)
    if rope_scaling:
        d["rope_scaling"] = rope_scaling
    d.update(kw)
    return d


_L31_ROPE = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                 original_max_position_embeddings=8192)
_L32_ROPE = dict(_L31_ROPE, factor=32.0)

PRESETS = {
    "meta-llama-3-8b": _llama(4096, 14336, 32, 32, 8),
    "meta-llama-3-8b-instruct": _llama(4096, 14336, 32, 32, 8),
    "llama-3-8b": _llama(4096, 14336, 32, 32, 8),
    "llama-3.1-8b": _llama(4096, 14336, 32, 32, 8, maxpos=131072, rope_scaling=_L31_ROPE),
    "llama-3.1-8b-instruct": _llama(4096, 14336, 32, 32, 8, maxpos=131072, rope_scaling=_L31_ROPE),
    "meta-llama-3-70b": _llama(8192, 28672, 80, 64, 8),
    "meta-llama-3-70b-instruct": _llama(8192, 28672, 80, 64, 8),
    "llama-3-70b": _llama(8192, 28672, 80, 64, 8),
    "llama-3.1-70b": _llama(8192, 28672, 80, 64, 8, maxpos=131072, rope_scaling=_L31_ROPE),
    "llama-3.2-1b": _llama(2048, 8192, 16, 32, 8, maxpos=131072, rope_scaling=_L32_ROPE, tie=True,
                           head_dim=64),
    "llama-3.2-3b": _llama(3072, 8192, 28, 24, 8, maxpos=131072, rope_scaling=_L32_ROPE, tie=True,
                           head_dim=128),
    "mistral-7b-v0.1": _llama(4096, 14336, 32, 32, 8, vocab=32000, theta=10000.0, maxpos=32768,
                              arch="MistralForCausalLM", model_type="mistral", sliding_window=4096,
                              bos_token_id=1, eos_token_id=2),
    "qwen2.5-7b": _llama(3584, 18944, 28, 28, 4, vocab=152064, theta=1000000.0, maxpos=32768,
                         arch="Qwen2ForCausalLM", model_type="qwen2", rms_norm_eps=1e-6,
                         bos_token_id=151643, eos_token_id=151643, attention_bias=True),
    "qwen3-8b": _llama(4096, 12288, 36, 32, 8, vocab=151936, theta=1000000.0, maxpos=40960,
                       arch="Qwen3ForCausalLM", model_type="qwen3", rms_norm_eps=1e-6, head_dim=128,
                       bos_token_id=151643, eos_token_id=151645),
    "mixtral-8x7b-v0.1": _llama(4096, 14336, 32, 32, 8, vocab=32000, theta=1000000.0, maxpos=32768,
                                arch="MixtralForCausalLM", model_type="mixtral", num_local_experts=8,
                                num_experts_per_tok=2, bos_token_id=1, eos_token_id=2),
    "mixtral-8x7b-instruct-v0.1": _llama(4096, 14336, 32, 32, 8, vocab=32000, theta=1000000.0,
                                         maxpos=32768, arch="MixtralForCausalLM", model_type="mixtral",
                                         num_local_experts=8, num_experts_per_tok=2,
                                         bos_token_id=1, eos_token_id=2),
    "qwen3-30b-a3b": _llama(2048, 6144, 48, 32, 4, vocab=151936, theta=1000000.0, maxpos=40960,
                            arch="Qwen3MoeForCausalLM", model_type="qwen3_moe", rms_norm_eps=1e-6,
                            head_dim=128, num_experts=128, num_experts_per_tok=8,
                            moe_intermediate_size=768, norm_topk_prob=True, decoder_sparse_step=1,
                            bos_token_id=151643, eos_token_id=151645),
    "opt-125m": dict(architectures=["OPTForCausalLM"], model_type="opt", hidden_size=768, ffn_dim=3072,
                     num_hidden_layers=12, num_attention_heads=12, vocab_size=50272,
                     max_position_embeddings=2048, word_embed_proj_dim=768, do_layer_norm_before=True,
                     activation_function="relu", enable_bias=True, layer_norm_elementwise_affine=True,
                     torch_dtype="float16", bos_token_id=2, eos_token_id=2, pad_token_id=1,
                     tie_word_embeddings=True),
    "gpt2": dict(architectures=["GPT2LMHeadModel"], model_type="gpt2", n_embd=768, n_layer=12, n_head=12,
                 vocab_size=50257, n_positions=1024, layer_norm_epsilon=1e-5,
                 activation_function="gelu_new", torch_dtype="float32", bos_token_id=50256,
                 eos_token_id=50256, tie_word_embeddings=True),
}
