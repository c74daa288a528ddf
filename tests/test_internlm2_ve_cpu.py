"""InternLM2 with visual experts under the InternVL wrapper (remote code upstream: parity with the original is
unpinned).  The routing is pinned by construction: (a) with ``feed_forward_ve`` equal to ``feed_forward`` every prompt
decodes exactly like the plain InternLM2 decoder under the same wrapper; (b) with different expert weights a text-only
prompt still decodes like the plain decoder, while a prompt with an image follows a plain PyTorch forward that sends
image rows through the second MLP."""
import json
import os

import numpy as np
import torch
from safetensors.torch import load_file, save_file

from vllm_distributed_amd import LLM, SamplingParams


def _internvl_checkpoint(tmp, name, llm_arch, ve_from=None):
    """InternVLChatModel layout over a tiny InternLM2-named decoder; ``ve_from``: None (no experts), "copy" or
    "random"."""
    import transformers as tf
    torch.manual_seed(0)
    vc = tf.InternVLVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                                 image_size=[28, 28], patch_size=[7, 7], use_qk_norm=False, attention_bias=True)
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                        tie_word_embeddings=False)
    hf = tf.InternVLForConditionalGeneration(tf.InternVLConfig(vision_config=vc, text_config=tc, image_token_id=300,
                                                               downsample_ratio=0.5)).float().eval()
    path = os.path.join(tmp, name)
    hf.save_pretrained(path, safe_serialization=True)
    sd = load_file(os.path.join(path, "model.safetensors"))
    H, KV, D = 2, 1, 64
    g = H // KV
    out, qkv, lm = {}, {}, {}
    ren = (("embeddings.cls_token", "embeddings.class_embedding"),
           ("embeddings.patch_embeddings.projection.", "embeddings.patch_embedding."),
           ("embeddings.position_embeddings", "embeddings.position_embedding"), ("encoder.layer.", "encoder.layers."),
           (".attention.projection_layer.", ".attn.proj."), (".lambda_1", ".ls1"), (".lambda_2", ".ls2"),
           (".layernorm_before.", ".norm1."), (".layernorm_after.", ".norm2."))
    lm_ren = (("embed_tokens", "tok_embeddings"), (".self_attn.o_proj.", ".attention.wo."),
              (".input_layernorm.", ".attention_norm."), (".post_attention_layernorm.", ".ffn_norm."),
              (".mlp.gate_proj.", ".feed_forward.w1."), (".mlp.up_proj.", ".feed_forward.w3."),
              (".mlp.down_proj.", ".feed_forward.w2."))
    for n, t in sd.items():
        if n.startswith("model.") and not n.startswith("model.language_model."):
            n = n[len("model."):]                      # older transformers layouts keep a "model." level
        if n.startswith("model.language_model."):
            n = "language_model.model." + n[len("model.language_model."):]
        if n.startswith("lm_head."):
            n = "language_model." + n
        if n.startswith("language_model.lm_head."):
            lm["output." + n[len("language_model.lm_head."):]] = t
        elif n.startswith("language_model."):
            lm[n[len("language_model."):]] = t
        elif n.startswith("multi_modal_projector."):
            n = n[len("multi_modal_projector."):]
            for a, b in (("layer_norm.", "0."), ("linear_1.", "1."), ("linear_2.", "3.")):
                if n.startswith(a):
                    n = b + n[len(a):]
            out["mlp1." + n] = t
        elif n.startswith("vision_tower."):
            n = n[len("vision_tower."):]
            for a, b in ren:
                n = n.replace(a, b)
            hit = next((s for s in ("q_proj", "k_proj", "v_proj") if f".attention.{s}." in n), None)
            if hit:
                qkv.setdefault(n.replace(f".attention.{hit}.", ".attn.qkv."), {})[hit] = t
            elif not n.startswith("layernorm."):
                out["vision_model." + n] = t
    for n, d in qkv.items():
        out["vision_model." + n] = torch.cat((d["q_proj"], d["k_proj"], d["v_proj"]), dim=0)
    gen = torch.Generator().manual_seed(9)
    for n, t in lm.items():
        if ".self_attn.q_proj." in n:
            k_, v_ = lm[n.replace("q_proj", "k_proj")], lm[n.replace("q_proj", "v_proj")]
            fused = torch.cat([t.view(KV, g, D, -1), k_.view(KV, 1, D, -1), v_.view(KV, 1, D, -1)], dim=1)
            out["language_model." + n.replace("self_attn.q_proj", "attention.wqkv")] = fused.reshape((H + 2 * KV) * D, -1)
            continue
        if ".k_proj." in n or ".v_proj." in n:
            continue
        m = n
        for a, b in lm_ren:
            m = m.replace(a, b)
        out["language_model." + m] = t
        if ve_from and ".feed_forward." in m:
            out["language_model." + m.replace(".feed_forward.", ".feed_forward_ve.")] = \
                t.clone() if ve_from == "copy" else torch.randn(t.shape, generator=gen) * 0.08
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(path, "model.safetensors"))
    c = json.load(open(os.path.join(path, "config.json")))
    new = {"architectures": ["InternVLChatModel"], "model_type": "internvl_chat", "downsample_ratio": 0.5,
           "select_layer": -1, "ps_version": "v2", "dynamic_image_size": True, "max_dynamic_patch": 12,
           "min_dynamic_patch": 1, "use_thumbnail": True,
           "llm_config": dict(c["text_config"], architectures=[llm_arch], model_type="internlm2"),
           "vision_config": {"hidden_size": 64, "intermediate_size": 128, "num_hidden_layers": 2,
                             "num_attention_heads": 2, "image_size": 28, "patch_size": 7, "qk_normalization": False,
                             "qkv_bias": True, "norm_type": c["vision_config"].get("norm_type", "layer_norm"),
                             "layer_norm_eps": c["vision_config"].get("layer_norm_eps", 1e-6), "hidden_act": "gelu"}}
    json.dump(new, open(os.path.join(path, "config.json"), "w"))
    return path


def _run(path):
    rng = np.random.default_rng(3)
    img = (rng.random((30, 60, 3)) * 255).astype(np.uint8)
    reqs = [{"prompt_token_ids": [5, 17, 300, 99, 3], "multi_modal_data": {"image": img}},
            {"prompt_token_ids": [7, 8, 9, 10, 11, 12, 13]}]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=8,
              max_num_batched_tokens=128, hf_overrides={"image_token_id": 300})
    outs = llm.generate(reqs, SamplingParams(temperature=0, max_tokens=6, ignore_eos=True, logprobs=0))
    names = {n for n, _ in llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
             .named_parameters()}
    llm.shutdown()
    return [(list(o.outputs[0].token_ids), [o.outputs[0].logprobs[k][t].logprob for k, t in
                                            enumerate(o.outputs[0].token_ids)]) for o in outs], names


def test_visual_experts_route_image_rows_only(tmp_path):
    plain, names = _run(_internvl_checkpoint(str(tmp_path), "plain", "InternLM2ForCausalLM"))
    assert not any("mlp_ve" in n for n in names)
    same, names = _run(_internvl_checkpoint(str(tmp_path), "copy", "InternLM2VEForCausalLM", "copy"))
    assert any("mlp_ve.gate_up_proj" in n for n in names)
    assert same == plain                                    # equal experts: nothing changes anywhere
    other, _ = _run(_internvl_checkpoint(str(tmp_path), "random", "InternLM2VEForCausalLM", "random"))
    assert other[1] == plain[1]                             # no image rows: the first MLP everywhere
    gap = max(abs(a - b) for a, b in zip(other[0][1], plain[0][1]))
    assert other[0][0] != plain[0][0] or gap > 1e-3         # image rows went through the second MLP
