"""Vision-language path: image preprocessing, placeholder expansion, hashing, and end-to-end LLaVA
generation equal to Hugging Face (modelled on tests/models/multimodal/generation and tests/multimodal)."""
import os

import numpy as np
import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.multimodal import ImageTokenProcessor, hash_item, to_pixel_values


def test_preprocessing_and_placeholders():
    img = (np.random.default_rng(0).random((50, 80, 3)) * 255).astype(np.uint8)
    px = to_pixel_values(img, 28)
    assert px.shape == (3, 28, 28) and px.dtype == torch.float32
    assert hash_item(img) == hash_item(img.copy()) and hash_item(img) != hash_item(img[::-1].copy())
    proc = ImageTokenProcessor(image_token_id=300, tokens_per_image=4, image_size=28)
    ids, inputs, hashes, pos = proc([1, 300, 2, 3, 300, 300, 300, 300, 9], {"image": [img, img[:, ::-1].copy()]})
    # a single marker and an already expanded run both become exactly tokens_per_image placeholders
    assert ids == [1, 300, 300, 300, 300, 2, 3, 300, 300, 300, 300, 9]
    assert [(p.offset, p.length) for p in pos] == [(1, 4), (7, 4)] and len(inputs) == 2 and hashes[0] != hashes[1]
    with pytest.raises(ValueError):
        proc([1, 2, 3], {"image": img})


def test_llava_matches_hf(tmp_path):
    import transformers as tf
    torch.manual_seed(0)
    vc = tf.CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                             image_size=28, patch_size=14, projection_dim=64)
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, head_dim=64)
    cfg = tf.LlavaConfig(vision_config=vc, text_config=tc, image_token_index=300, vision_feature_layer=-2,
                         vision_feature_select_strategy="default")
    hf = tf.LlavaForConditionalGeneration(cfg).float().eval()
    path = os.path.join(tmp_path, "llava")
    hf.save_pretrained(path, safe_serialization=True)
    rng = np.random.default_rng(1)
    imgs = [(rng.random((28, 28, 3)) * 255).astype(np.uint8) for _ in range(2)]
    prompts = [[5, 17, 300, 99, 3], [300, 7, 8, 9, 10, 11, 12], [1, 2, 3, 4]]
    mm = [{"image": imgs[0]}, {"image": imgs[1]}, None]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64)
    reqs = [{"prompt_token_ids": p, **({"multi_modal_data": m} if m else {})} for p, m in zip(prompts, mm)]
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
    outs = llm.generate(reqs, sp)
    # the same image again: served from the prefix cache, tokens unchanged
    again = llm.generate(reqs[:1], sp)
    llm.shutdown()
    n_img = (28 // 14) ** 2
    for p, m, o in zip(prompts, mm, outs):
        ids = []
        for t in p:
            ids.extend([300] * n_img if t == 300 else [t])
        kw = {}
        if m:
            from vllm_distributed_amd.multimodal import to_pixel_values as tpv
            kw["pixel_values"] = tpv(m["image"], 28).unsqueeze(0)
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None, **kw)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, p
    assert list(again[0].outputs[0].token_ids) == list(outs[0].outputs[0].token_ids)
