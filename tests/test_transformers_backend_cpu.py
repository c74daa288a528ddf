"""The transformers fallback (models/transformers_backend.py): transformers' own module code around this
package's paged attention and KV cache must decode the same greedy tokens as transformers' generate(),
for an architecture this package implements natively (forced with model_impl="transformers") and for
architectures it does not carry (resolved automatically).  Modelled on the reference's
tests/models/test_transformers.py."""
import os

import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.models.registry import ModelRegistry


def _tiny(tmp, kind):
    import transformers as tf
    torch.manual_seed(0)
    kw = dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
              num_key_value_heads=1, max_position_embeddings=256, head_dim=64)
    if kind == "llama":
        m = tf.LlamaForCausalLM(tf.LlamaConfig(tie_word_embeddings=False, **kw))
    elif kind == "helium":
        m = tf.HeliumForCausalLM(tf.HeliumConfig(pad_token_id=0, **kw))
    elif kind == "smollm3":          # every third layer without rotary positions
        m = tf.SmolLM3ForCausalLM(tf.SmolLM3Config(pad_token_id=0, no_rope_layer_interval=3, **kw))
    elif kind == "gemma2":           # sliding-window layers, soft-capped scores and logits, query scaling
        m = tf.Gemma2ForCausalLM(tf.Gemma2Config(sliding_window=16, query_pre_attn_scalar=32, pad_token_id=0, **kw))
    m = m.float().eval()
    path = os.path.join(tmp, kind)
    m.save_pretrained(path, safe_serialization=True)
    return path, m


@pytest.mark.parametrize("kind,impl", [("llama", "transformers"), ("helium", "auto"), ("smollm3", "auto"),
                                       ("gemma2", "transformers")])
def test_transformers_fallback_matches_hf_generate(tmp_path, kind, impl):
    path, hf = _tiny(str(tmp_path), kind)
    arch = hf.config.architectures[0]
    if impl == "auto":
        assert not ModelRegistry.is_supported(arch), f"{arch} has a native implementation now: pick another"
    prompts = [[5, 17, 99, 3, 250, 11], [7] * 40, [1, 2, 3]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64, enable_prefix_caching=False, block_size=16, model_impl=impl)
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    assert type(model).__name__ == "TransformersForCausalLM"
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=12, ignore_eos=True))
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf.generate(torch.tensor([p]), max_new_tokens=12, do_sample=False, pad_token_id=0,
                              eos_token_id=None)[0, len(p):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (kind, p)
    llm.shutdown()


def test_model_impl_vllm_refuses_unknown_architecture(tmp_path):
    path, _ = _tiny(str(tmp_path), "helium")
    with pytest.raises(Exception, match="not supported"):
        LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=64, model_impl="vllm")
