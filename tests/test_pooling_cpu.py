"""Pooling models: last-token embeddings and sequence classification equal the HF reference
(modelled on tests/models/language/pooling)."""
import os

import torch

from vllm_distributed_amd import LLM

PROMPTS = [[5, 17, 99, 3, 250, 11], [7] * 30, [1, 2, 3]]
KW = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
          max_num_batched_tokens=256)


def test_embedding_and_classification_match_hf(tmp_path):
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, head_dim=64,
                         num_labels=3, pad_token_id=0)
    emb_hf = tf.LlamaModel(cfg).float().eval()
    p1 = os.path.join(tmp_path, "emb")
    emb_hf.save_pretrained(p1, safe_serialization=True)
    llm = LLM(model=p1, task="embed", **KW)
    outs = llm.embed([{"prompt_token_ids": p} for p in PROMPTS])
    llm.shutdown()
    for p, o in zip(PROMPTS, outs):
        with torch.no_grad():
            h = emb_hf(torch.tensor([p])).last_hidden_state[0, -1]
        ref = torch.nn.functional.normalize(h, dim=-1)
        torch.testing.assert_close(torch.tensor(o.outputs.embedding), ref, atol=1e-4, rtol=1e-3)

    cls_hf = tf.LlamaForSequenceClassification(cfg).float().eval()
    p2 = os.path.join(tmp_path, "cls")
    cls_hf.save_pretrained(p2, safe_serialization=True)
    llm = LLM(model=p2, task="classify", **KW)
    outs = llm.classify([{"prompt_token_ids": p} for p in PROMPTS])
    llm.shutdown()
    for p, o in zip(PROMPTS, outs):
        with torch.no_grad():
            ref = torch.softmax(cls_hf(torch.tensor([p])).logits[0], dim=-1)
        torch.testing.assert_close(torch.tensor(o.outputs.probs), ref, atol=1e-4, rtol=1e-3)
