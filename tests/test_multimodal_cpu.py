"""Vision-language path: image preprocessing, placeholder expansion, hashing, and end-to-end LLaVA
generation equal to Hugging Face (modelled on tests/models/multimodal/generation and tests/multimodal)."""
import os

import numpy as np
import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.multimodal import ImageTokenProcessor, hash_item, to_pixel_values


def _to(model, x):
    """Tensors (nested in lists) moved to where the model under test lives: these tests also run with the engines on
    the GPU (tests/conftest.py, VDA_TEST_DEVICE)."""
    if isinstance(x, (list, tuple)):
        return [_to(model, v) for v in x]
    return x.to(next(model.parameters()).device) if isinstance(x, torch.Tensor) else x


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


def _tiny_qwen2_vl(tmp):
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.Qwen2VLConfig(
        text_config=dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                         rope_parameters={"rope_type": "default", "mrope_section": [8, 12, 12], "rope_theta": 10000.0},
                         bos_token_id=1, eos_token_id=2),
        vision_config=dict(depth=2, embed_dim=64, hidden_size=128, num_heads=2, mlp_ratio=2, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3),
        image_token_id=300, video_token_id=301, vision_start_token_id=302, vision_end_token_id=303)
    m = tf.Qwen2VLForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)
    m = m.float().eval()
    path = os.path.join(tmp, "qwen2vl")
    m.save_pretrained(path, safe_serialization=True)
    return path, m


def test_mrope_positions_layout():
    from vllm_distributed_amd.models.qwen2_vl import mrope_positions
    # 3 text tokens, an image of 1 x 4 x 6 patches (2 x 3 after merging), 2 text tokens
    pos, delta = mrope_positions(3 + 6 + 2, [(3, 6)], [(1, 4, 6)], 2)
    assert pos[:, :3].tolist() == [[0, 1, 2]] * 3
    assert pos[0, 3:9].tolist() == [3] * 6
    assert pos[1, 3:9].tolist() == [3, 3, 3, 4, 4, 4]
    assert pos[2, 3:9].tolist() == [3, 4, 5, 3, 4, 5]
    assert pos[:, 9:].tolist() == [[6, 7]] * 3           # continues after max(t, h, w) = 3
    assert delta == 8 - 11


def test_qwen2_vl_matches_hf(tmp_path):
    """Vision tower over two different patch grids, merged embeddings, multimodal rotary positions in
    prefill (chunked) and the position offset in decode: greedy tokens equal transformers'."""
    path, hf = _tiny_qwen2_vl(str(tmp_path))
    g = torch.Generator().manual_seed(3)
    imgs = [{"pixel_values": torch.randn(4 * 6, 1176, generator=g), "image_grid_thw": torch.tensor([1, 4, 6])},
            {"pixel_values": torch.randn(8 * 4, 1176, generator=g), "image_grid_thw": torch.tensor([1, 8, 4])}]
    vid = {"pixel_values_videos": torch.randn(3 * 4 * 4, 1176, generator=g), "video_grid_thw": torch.tensor([3, 4, 4])}
    prompts = [([5, 17, 302, 300, 303, 9, 11, 250], [imgs[0]]),
               ([7, 8, 9, 10] * 5, []),
               ([3, 302, 300, 303, 44, 45, 302, 300, 303, 46], [imgs[1], imgs[0]]),
               ([9, 302, 301, 303, 12, 13], [vid])]
    reqs = [{"prompt_token_ids": p, **({"multi_modal_data": {"video" if "video_grid_thw" in m[0] else "image": m}}
                                        if m else {})} for p, m in prompts]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=16, enable_prefix_caching=False, block_size=16)
    outs = llm.generate(reqs, SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    for (p, ims), o in zip(prompts, outs):
        ids, k = [], 0
        for t in p:
            if t in (300, 301):
                n = int(ims[k]["image_grid_thw" if t == 300 else "video_grid_thw"].prod()) // 4
                ids += [t] * n
                k += 1
            else:
                ids.append(t)
        assert list(o.prompt_token_ids) == ids
        kw = {}
        if ims and "video_grid_thw" in ims[0]:
            kw = dict(pixel_values_videos=ims[0]["pixel_values_videos"], video_grid_thw=ims[0]["video_grid_thw"][None],
                      mm_token_type_ids=torch.tensor([[2 if t == 301 else 0 for t in ids]], dtype=torch.int32))
        elif ims:
            kw = dict(pixel_values=torch.cat([i["pixel_values"] for i in ims]),
                      image_grid_thw=torch.stack([i["image_grid_thw"] for i in ims]),
                      mm_token_type_ids=torch.tensor([[1 if t == 300 else 0 for t in ids]], dtype=torch.int32))
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                              eos_token_id=None, **kw)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (p, list(o.outputs[0].token_ids), ref)
    llm.shutdown()


def test_patch_grid_preprocessing_matches_hf_layout():
    """Same grid and patch order as transformers' Qwen2-VL image processor; pixel values differ only by
    the resize filter (PIL on uint8 there, float bicubic here)."""
    np = pytest.importorskip("numpy")
    pytest.importorskip("PIL")
    from PIL import Image
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    from vllm_distributed_amd.multimodal import PatchGridProcessor, smart_resize
    img = (np.random.default_rng(0).random((120, 200, 3)) * 255).astype(np.uint8)
    px, grid = PatchGridProcessor(300, 14, 2, 2, 56 * 56, 14 * 14 * 4 * 1280).patches(img)
    ref = Qwen2VLImageProcessorPil(min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280)(
        images=[Image.fromarray(img)], return_tensors="pt")
    assert grid.tolist() == ref["image_grid_thw"][0].tolist() == [1, 8, 14]
    assert px.shape == ref["pixel_values"].shape
    assert (px - ref["pixel_values"]).abs().mean() < 0.02
    assert smart_resize(120, 200, 28) == (112, 196)


def test_video_frames_to_patch_grid():
    from vllm_distributed_amd.multimodal import PatchGridProcessor
    import numpy as np
    proc = PatchGridProcessor(300, 14, 2, 2, 56 * 56, 14 * 14 * 4 * 1280, video_token_id=301)
    frames = (np.random.default_rng(1).random((3, 56, 84, 3)) * 255).astype(np.uint8)
    px, grid = proc.video_patches(frames)
    assert grid.tolist() == [2, 4, 6] and px.shape == (2 * 4 * 6, 1176)      # odd frame count: last frame repeated
    ids, inputs, hashes, pos = proc([1, 301, 2, 300, 3], {"video": frames, "image": frames[0]})
    assert [p.length for p in pos] == [12, 6] and ids.count(301) == 12 and ids.count(300) == 6
    # a still image is the video of its frame shown twice
    one, g1 = proc.patches(frames[0])
    two, g2 = proc.video_patches(np.stack([frames[0], frames[0]]))
    assert g1.tolist() == g2.tolist() and torch.equal(one, two)


def test_qwen2_5_vl_matches_hf(tmp_path):
    """Windowed vision attention (window of 2 x 2 merge groups, one full-attention block), RMSNorm / gated
    MLP tower, window reordering undone after the merger."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.Qwen2_5_VLConfig(
        text_config=dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                         rope_parameters={"rope_type": "default", "mrope_section": [8, 12, 12], "rope_theta": 10000.0},
                         bos_token_id=1, eos_token_id=2),
        vision_config=dict(depth=3, hidden_size=64, intermediate_size=96, num_heads=2, out_hidden_size=128,
                           patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=56,
                           fullatt_block_indexes=[1], in_channels=3),
        image_token_id=300, video_token_id=301, vision_start_token_id=302, vision_end_token_id=303)
    hf = tf.Qwen2_5_VLForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)
    hf = hf.float().eval()
    path = os.path.join(str(tmp_path), "qwen25vl")
    hf.save_pretrained(path, safe_serialization=True)
    g = torch.Generator().manual_seed(5)
    img = {"pixel_values": torch.randn(6 * 10, 1176, generator=g), "image_grid_thw": torch.tensor([1, 6, 10])}
    p = [5, 17, 302, 300, 303, 9, 11, 250]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64, enable_prefix_caching=False, block_size=16)
    o = llm.generate([{"prompt_token_ids": p, "multi_modal_data": {"image": [img]}}],
                     SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
    ids = [5, 17, 302] + [300] * 15 + [303, 9, 11, 250]
    assert list(o.prompt_token_ids) == ids
    with torch.no_grad():
        ref = hf.generate(input_ids=torch.tensor([ids]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                          eos_token_id=None, pixel_values=img["pixel_values"], image_grid_thw=img["image_grid_thw"][None],
                          mm_token_type_ids=torch.tensor([[1 if t == 300 else 0 for t in ids]], dtype=torch.int32)
                          )[0, len(ids):].tolist()
    assert list(o.outputs[0].token_ids) == ref
    # a video: frames are positioned by their distance in time (seconds per temporal step x tokens per second)
    vid = {"pixel_values_videos": torch.randn(3 * 4 * 4, 1176, generator=g), "video_grid_thw": torch.tensor([3, 4, 4]),
           "second_per_grid_ts": 1.5}
    pv = [5, 302, 301, 303, 9, 11]
    o = llm.generate([{"prompt_token_ids": pv, "multi_modal_data": {"video": vid}}],
                     SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
    ids = [5, 302] + [301] * 12 + [303, 9, 11]
    assert list(o.prompt_token_ids) == ids
    with torch.no_grad():
        ref = hf.generate(input_ids=torch.tensor([ids]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                          eos_token_id=None, pixel_values_videos=vid["pixel_values_videos"],
                          video_grid_thw=vid["video_grid_thw"][None], second_per_grid_ts=torch.tensor([1.5]),
                          mm_token_type_ids=torch.tensor([[2 if t == 301 else 0 for t in ids]], dtype=torch.int32)
                          )[0, len(ids):].tolist()
    assert list(o.outputs[0].token_ids) == ref
    from vllm_distributed_amd.models.qwen2_vl import mrope_positions
    pos, delta = mrope_positions(len(ids), [(2, 12)], [(3, 4, 4)], 2, [1.5 * 2])
    assert pos[0, 2:14].tolist() == [2] * 4 + [5] * 4 + [8] * 4 and pos[0, 14] == 9 and delta == 9 - 14
    llm.shutdown()


def test_qwen2_audio_matches_hf(tmp_path):
    """Audio clips of different lengths through the Whisper-style encoder (padded frames masked), pooled and
    projected into a Qwen2 decoder: greedy tokens equal transformers'."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.Qwen2AudioConfig(
        audio_config=dict(num_mel_bins=128, d_model=64, encoder_layers=2, encoder_attention_heads=2,
                          encoder_ffn_dim=128, max_source_positions=1500),
        text_config=dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=1024,
                         tie_word_embeddings=False),
        audio_token_index=300)
    hf = tf.Qwen2AudioForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2 and "audio_tower" in n:
                p.mul_(3.0)                      # make the clip matter for the next tokens
    hf = hf.float().eval()
    path = os.path.join(tmp_path, "qwen2audio")
    hf.save_pretrained(path, safe_serialization=True)
    rng = np.random.default_rng(3)
    clips = [rng.standard_normal(16000 * 2).astype(np.float32) * 0.1, rng.standard_normal(9000).astype(np.float32) * 0.1]
    prompts = [[5, 17, 300, 99, 3], [300, 7, 8, 9], [1, 2, 3, 4]]
    mm = [{"audio": clips[0]}, {"audio": (clips[1], 16000)}, None]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=8,
              max_num_batched_tokens=256)
    reqs = [{"prompt_token_ids": p, **({"multi_modal_data": m} if m else {})} for p, m in zip(prompts, mm)]
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
    outs = llm.generate(reqs, sp)
    llm.shutdown()
    from vllm_distributed_amd.multimodal import AudioTokenProcessor
    proc = AudioTokenProcessor(300, 128)
    for p, m, o in zip(prompts, mm, outs):
        kw, ids = {}, list(p)
        if m:
            feats, mel = proc.features(m["audio"])
            n = proc.num_tokens(mel)
            ids = []
            for t in p:
                ids.extend([300] * n if t == 300 else [t])
            mask = torch.zeros(1, feats.shape[-1], dtype=torch.long)
            mask[0, :mel] = 1
            kw = dict(input_features=feats.unsqueeze(0), feature_attention_mask=mask)
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), attention_mask=torch.ones(1, len(ids), dtype=torch.long),
                              max_new_tokens=8, do_sample=False, pad_token_id=0, eos_token_id=None,
                              **kw)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, p


def test_chat_audio_parts_decode_to_waveforms():
    """OpenAI ``input_audio`` and ``audio_url`` content parts: WAV bytes become (waveform, rate) clips and leave
    an audio placeholder in the text."""
    import base64
    import io

    from scipy.io import wavfile

    from vllm_distributed_amd.entrypoints.chat_utils import load_media, parse_chat_messages
    buf = io.BytesIO()
    wavfile.write(buf, 8000, (np.sin(np.arange(8000) / 10) * 20000).astype(np.int16))
    b64 = base64.b64encode(buf.getvalue()).decode()
    msgs = [{"role": "user", "content": [
        {"type": "text", "text": "what is this"},
        {"type": "input_audio", "input_audio": {"data": b64, "format": "wav"}},
        {"type": "audio_url", "audio_url": {"url": "data:audio/wav;base64," + b64}}]}]
    conv, media = parse_chat_messages(msgs)
    assert conv[0]["content"].count("<audio>") == 2
    mm = load_media(media)
    assert len(mm["audio"]) == 2
    wave, rate = mm["audio"][0]
    assert rate == 8000 and wave.shape == (8000,) and 0.5 < float(np.abs(wave).max()) <= 1.0
    # another sampling rate is resampled by the audio processor
    from vllm_distributed_amd.multimodal import AudioTokenProcessor
    proc = AudioTokenProcessor(300, 128)
    feats, mel = proc.features((wave, rate))
    assert feats.shape == (128, 3000) and mel == 100          # 1 s of audio = 100 mel frames


def test_whisper_matches_hf(tmp_path):
    """Encoder-decoder: the audio encoder output reaches the decoder through the paged cross-attention cache
    (written with the first decoder step, read at every later one).  Greedy tokens equal transformers'."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.WhisperConfig(vocab_size=320, num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2,
                           decoder_layers=2, decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256,
                           max_source_positions=1500, max_target_positions=64, pad_token_id=0, bos_token_id=1,
                           eos_token_id=2, decoder_start_token_id=3, suppress_tokens=None, begin_suppress_tokens=None)
    hf = tf.WhisperForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)                      # let the audio decide the tokens
    hf = hf.float().eval()
    hf.generation_config = tf.GenerationConfig(pad_token_id=0, eos_token_id=None, decoder_start_token_id=3)
    path = os.path.join(tmp_path, "whisper")
    hf.save_pretrained(path, safe_serialization=True)
    rng = np.random.default_rng(5)
    clips = [rng.standard_normal(16000 * 3).astype(np.float32) * 0.2, rng.standard_normal(20000).astype(np.float32) * 0.2,
             rng.standard_normal(5000).astype(np.float32) * 0.2]
    dec_prompts = [[3, 10, 11, 12], [3, 10], [3, 20, 21, 22, 23, 24]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=64, max_num_seqs=8,
              max_num_batched_tokens=64)
    reqs = [{"prompt_token_ids": p, "multi_modal_data": {"audio": c}} for p, c in zip(dec_prompts, clips)]
    sp = SamplingParams(temperature=0, max_tokens=10, ignore_eos=True)
    outs = llm.generate(reqs, sp)
    # a second wave reuses batch slots: their cross-attention caches must be rewritten
    outs2 = llm.generate([reqs[2], reqs[0]], sp)
    llm.shutdown()
    from vllm_distributed_amd.multimodal import AudioTokenProcessor
    proc = AudioTokenProcessor(-1, 80)
    refs = []
    for p, c in zip(dec_prompts, clips):
        feats, _ = proc.features(c)
        ids, ref = list(p), []
        with torch.no_grad():                       # plain greedy loop (Whisper's generate() post-processes its output)
            enc = hf.model.encoder(feats.unsqueeze(0)).last_hidden_state
            for _ in range(10):
                lg = hf(encoder_outputs=(enc,), decoder_input_ids=torch.tensor([ids])).logits[0, -1]
                ref.append(int(lg.argmax()))
                ids.append(ref[-1])
        refs.append(ref)
    for o, ref in zip(outs, refs):
        assert list(o.outputs[0].token_ids) == ref
    assert list(outs2[0].outputs[0].token_ids) == refs[2] and list(outs2[1].outputs[0].token_ids) == refs[0]


@pytest.mark.parametrize("family", ["next", "onevision"])
def test_llava_next_and_onevision_match_hf(tmp_path, family):
    """Any-resolution images: tiles + base image through the tower, the tile map unpadded (and, for OneVision, capped)
    with a newline vector per row.  Images of three aspect ratios, so that padding is cut from rows and from columns;
    the processor's own tiles are fed to transformers as well (its image processor needs torchvision)."""
    import transformers as tf
    from vllm_distributed_amd.models.llava_next import AnyResImageProcessor, anyres_num_tokens
    torch.manual_seed(0)
    pins = [[28, 28], [28, 56], [56, 28], [56, 56], [28, 84]]
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, head_dim=64)
    if family == "next":
        vc = tf.CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                                 image_size=28, patch_size=7, projection_dim=64)
        cfg = tf.LlavaNextConfig(vision_config=vc, text_config=tc, image_token_index=300, vision_feature_layer=-2,
                                 vision_feature_select_strategy="default", image_grid_pinpoints=pins)
        hf = tf.LlavaNextForConditionalGeneration(cfg).float().eval()
        max_tiles = 0
    else:
        vc = tf.SiglipVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                                   image_size=28, patch_size=7)
        cfg = tf.LlavaOnevisionConfig(vision_config=vc, text_config=tc, image_token_index=300, vision_feature_layer=-1,
                                      vision_feature_select_strategy="full", image_grid_pinpoints=pins,
                                      vision_aspect_ratio="anyres_max_2")
        hf = tf.LlavaOnevisionForConditionalGeneration(cfg).float().eval()
        max_tiles = 2
    with torch.no_grad():
        hf.model.image_newline.normal_(0, 0.5)
    path = os.path.join(tmp_path, family)
    hf.save_pretrained(path, safe_serialization=True)
    rng = np.random.default_rng(1)
    imgs = [(rng.random(s + (3,)) * 255).astype(np.uint8) for s in ((40, 90), (100, 35), (50, 52))]
    prompts = [[5, 17, 300, 99, 3], [300, 7, 8, 9, 10, 11, 12], [4, 300, 6]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=8,
              max_num_batched_tokens=128)
    reqs = [{"prompt_token_ids": p, "multi_modal_data": {"image": im}} for p, im in zip(prompts, imgs)]
    sp = SamplingParams(temperature=0, max_tokens=6, ignore_eos=True)
    outs = llm.generate(reqs, sp)
    llm.shutdown()
    proc = AnyResImageProcessor(300, 28, 7, pins, (0.5,) * 3, (0.5,) * 3, max_tiles) if family == "onevision" else None
    if proc is None:
        from vllm_distributed_amd.multimodal import CLIP_MEAN, CLIP_STD
        proc = AnyResImageProcessor(300, 28, 7, pins, CLIP_MEAN, CLIP_STD, 0)
    lens = set()
    for p, im, o in zip(prompts, imgs, outs):
        tiles, size = proc.tiles(im)
        n_img = anyres_num_tokens((int(size[0]), int(size[1])), pins, 28, 7, max_tiles)
        lens.add(n_img)
        ids = []
        for t in p:
            ids.extend([300] * n_img if t == 300 else [t])
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=tiles[None], image_sizes=size[None],
                              max_new_tokens=6, do_sample=False, pad_token_id=0, eos_token_id=None)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (family, im.shape)
    assert len(lens) == 3                       # three different token counts: the geometry was really exercised


def test_fuyu_matches_hf(tmp_path):
    """Patches projected straight into the decoder's input, rows closed by the newline token."""
    import transformers as tf
    from vllm_distributed_amd.models.fuyu import IMAGE_TOKEN_ID, NEWLINE_TOKEN_ID, FuyuImageProcessor
    torch.manual_seed(0)
    tc = dict(vocab_size=71100, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
              max_position_embeddings=256, qk_layernorm=True, model_type="persimmon")
    cfg = tf.FuyuConfig(vocab_size=71100, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, max_position_embeddings=256, patch_size=6, text_config=tc)
    hf = tf.FuyuForCausalLM(cfg).float().eval()
    path = os.path.join(tmp_path, "fuyu")
    hf.save_pretrained(path, safe_serialization=True)
    rng = np.random.default_rng(1)
    imgs = [(rng.random(s + (3,)) * 255).astype(np.uint8) for s in ((13, 20), (6, 31))]
    prompts = [[5, 17, IMAGE_TOKEN_ID, 99, 3], [IMAGE_TOKEN_ID, 7, 8, 9]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=128)
    sp = SamplingParams(temperature=0, max_tokens=6, ignore_eos=True)
    outs = llm.generate([{"prompt_token_ids": p, "multi_modal_data": {"image": im}} for p, im in zip(prompts, imgs)], sp)
    llm.shutdown()
    proc = FuyuImageProcessor(6)
    for p, im, o in zip(prompts, imgs, outs):
        patches, grid = proc.patches(im)
        rows, cols = int(grid[0]), int(grid[1])
        ids, idx, k = [], [], 0
        for t in p:
            if t == IMAGE_TOKEN_ID:
                for _ in range(rows):
                    ids += [IMAGE_TOKEN_ID] * cols + [NEWLINE_TOKEN_ID]
                    idx += list(range(k, k + cols)) + [-1]
                    k += cols
            else:
                ids.append(t)
                idx.append(-1)
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), image_patches=patches[None],
                              image_patches_indices=torch.tensor([idx]), max_new_tokens=6, do_sample=False,
                              pad_token_id=0, eos_token_id=None)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, im.shape


@pytest.mark.parametrize("family", ["mistral3", "pixtral_llava"])
def test_pixtral_tower_models_match_hf(tmp_path, family):
    """Pixtral tower (images of any size, 2-D rotary positions, block-diagonal attention) under Mistral-3's
    norm + 2x2 patch merger + projector and under LLaVA's plain projector; image rows closed by break / end tokens
    that keep their text embeddings.  Three image shapes; the processor's own pixels are fed to transformers too."""
    import transformers as tf
    from vllm_distributed_amd.models.pixtral import PixtralImageProcessor
    from vllm_distributed_amd.multimodal import CLIP_MEAN, CLIP_STD
    torch.manual_seed(0)
    vc = tf.PixtralVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                                image_size=64, patch_size=4, head_dim=32)
    tc = tf.MistralConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                          num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=1024, head_dim=64,
                          sliding_window=None)
    if family == "mistral3":
        cfg = tf.Mistral3Config(vision_config=vc, text_config=tc, image_token_index=300, spatial_merge_size=2,
                                vision_feature_layer=-1)
        hf = tf.Mistral3ForConditionalGeneration(cfg).float().eval()
        merge, arch = 2, None
    else:
        cfg = tf.LlavaConfig(vision_config=vc, text_config=tc, image_token_index=300, vision_feature_layer=-1,
                             vision_feature_select_strategy="full")
        hf = tf.LlavaForConditionalGeneration(cfg).float().eval()
        merge, arch = 1, None           # resolved from the vision tower's model type under LLaVA's own name
    path = os.path.join(tmp_path, family)
    hf.save_pretrained(path, safe_serialization=True)
    overrides = {"image_break_token_id": 301, "image_end_token_id": 302}
    if arch:
        overrides["architectures"] = [arch]
    rng = np.random.default_rng(2)
    imgs = [(rng.random(s + (3,)) * 255).astype(np.uint8) for s in ((20, 41), (100, 30), (16, 16))]
    prompts = [[5, 17, 300, 99, 3], [300, 7, 8, 9, 10, 11, 12], [4, 300, 6]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=1024, max_num_seqs=8,
              max_num_batched_tokens=256, hf_overrides=overrides)
    reqs = [{"prompt_token_ids": p, "multi_modal_data": {"image": im}} for p, im in zip(prompts, imgs)]
    outs = llm.generate(reqs, SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    llm.shutdown()
    proc = PixtralImageProcessor(300, 301, 302, 4, merge, 64, CLIP_MEAN, CLIP_STD)
    shapes = set()
    for p, im, o in zip(prompts, imgs, outs):
        ids, inputs, _, ranges = proc(p, {"image": im})
        px = inputs[0][0]
        shapes.add(tuple(px.shape))
        assert ranges[0].length == (px.shape[1] // (4 * merge)) * (px.shape[2] // (4 * merge) + 1)
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=px[None],
                              image_sizes=torch.tensor([list(px.shape[1:])]), max_new_tokens=6, do_sample=False,
                              pad_token_id=0, eos_token_id=None)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (family, im.shape)
    assert len(shapes) == 3


@pytest.mark.parametrize("layout", ["transformers", "original"])
def test_internvl_matches_hf(tmp_path, layout):
    """InternViT tower (layer scale, query / key norm) + pixel shuffle + projector over a Qwen2 decoder, dynamic tiling
    with a thumbnail.  ``original``: the same weights re-laid-out the way the InternVLChatModel checkpoints store
    them (fused qkv, mlp1.*, language_model.*, llm_config) must load and decode identically."""
    import json
    import transformers as tf
    from safetensors.torch import load_file, save_file
    from vllm_distributed_amd.models.internvl import InternVLImageProcessor
    torch.manual_seed(0)
    vc = tf.InternVLVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                                 image_size=[28, 28], patch_size=[7, 7], use_qk_norm=True, attention_bias=True,
                                 layer_scale_init_value=0.5)
    tc = tf.Qwen2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                        tie_word_embeddings=False)
    cfg = tf.InternVLConfig(vision_config=vc, text_config=tc, image_token_id=300, downsample_ratio=0.5)
    hf = tf.InternVLForConditionalGeneration(cfg).float().eval()
    with torch.no_grad():
        for n, p_ in hf.named_parameters():
            if "lambda_" in n or "position_embeddings" in n or "cls_token" in n:
                p_.normal_(0, 0.5)
    path = os.path.join(tmp_path, layout)
    hf.save_pretrained(path, safe_serialization=True)
    overrides = {}
    if layout == "original":
        sd = load_file(os.path.join(path, "model.safetensors"))
        out, qkv = {}, {}
        ren = (("embeddings.cls_token", "embeddings.class_embedding"),
               ("embeddings.patch_embeddings.projection.", "embeddings.patch_embedding."),
               ("embeddings.position_embeddings", "embeddings.position_embedding"), ("encoder.layer.", "encoder.layers."),
               (".attention.projection_layer.", ".attn.proj."), (".attention.q_norm.", ".attn.q_norm."),
               (".attention.k_norm.", ".attn.k_norm."), (".lambda_1", ".ls1"), (".lambda_2", ".ls2"),
               (".layernorm_before.", ".norm1."), (".layernorm_after.", ".norm2."))
        for n, t in sd.items():
            if n.startswith("model.") and not n.startswith("model.language_model."):
                n = n[6:]
            if n.startswith("language_model."):
                out[n] = t
            elif n.startswith("model.language_model."):
                out["language_model.model." + n[len("model.language_model."):]] = t
            elif n.startswith("lm_head."):
                out["language_model." + n] = t
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
        save_file(out, os.path.join(path, "model.safetensors"))
        c = json.load(open(os.path.join(path, "config.json")))
        v = c["vision_config"]
        new = {"architectures": ["InternVLChatModel"], "model_type": "internvl_chat", "downsample_ratio": 0.5,
               "select_layer": -1, "ps_version": "v2", "dynamic_image_size": True, "max_dynamic_patch": 12,
               "min_dynamic_patch": 1, "use_thumbnail": True,
               "llm_config": dict(c["text_config"], architectures=["Qwen2ForCausalLM"]),
               "vision_config": {"hidden_size": 64, "intermediate_size": 128, "num_hidden_layers": 3,
                                 "num_attention_heads": 2, "image_size": 28, "patch_size": 7, "qk_normalization": True,
                                 "qkv_bias": True, "norm_type": v.get("norm_type", "layer_norm"),
                                 "layer_norm_eps": v.get("layer_norm_eps", 1e-6), "hidden_act": "gelu"}}
        json.dump(new, open(os.path.join(path, "config.json"), "w"))
        overrides = {"image_token_id": 300}
    rng = np.random.default_rng(3)
    imgs = [(rng.random(s + (3,)) * 255).astype(np.uint8) for s in ((30, 95), (100, 35), (28, 28))]
    prompts = [[5, 17, 300, 99, 3], [300, 7, 8, 9, 10, 11, 12], [4, 300, 6]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=8,
              max_num_batched_tokens=128, hf_overrides=overrides)
    reqs = [{"prompt_token_ids": p, "multi_modal_data": {"image": im}} for p, im in zip(prompts, imgs)]
    outs = llm.generate(reqs, SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    llm.shutdown()
    proc = InternVLImageProcessor(300, 28, 4, 1, 12, True, True)
    counts = set()
    for p, im, o in zip(prompts, imgs, outs):
        ids, inputs, _, _ = proc(p, {"image": im})
        tiles = inputs[0][0]
        counts.add(tiles.shape[0])
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=tiles, max_new_tokens=6, do_sample=False,
                              pad_token_id=0, eos_token_id=None)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, (layout, im.shape)
    assert counts == {4, 1}            # 3 x 1 and 1 x 3 grids + thumbnail, and a single tile


def test_llava_next_video_matches_hf(tmp_path):
    """Frames through the tower, stride-2 average pooling of every frame's feature map, projector; one request with
    a video, one with an image (the LLaVA-NeXT path of the same checkpoint), one with both."""
    import transformers as tf
    from vllm_distributed_amd.models.llava_next import FrameVideoProcessor
    from vllm_distributed_amd.multimodal import CLIP_MEAN, CLIP_STD
    torch.manual_seed(0)
    pins = [[28, 28], [28, 56], [56, 28]]
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, head_dim=64)
    vc = tf.CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                             image_size=28, patch_size=7, projection_dim=64)
    cfg = tf.LlavaNextVideoConfig(vision_config=vc, text_config=tc, image_token_index=300, video_token_index=301,
                                  vision_feature_layer=-2, vision_feature_select_strategy="default",
                                  image_grid_pinpoints=pins, spatial_pool_mode="average", spatial_pool_stride=2)
    hf = tf.LlavaNextVideoForConditionalGeneration(cfg).float().eval()
    with torch.no_grad():
        hf.model.image_newline.normal_(0, 0.5)
    path = os.path.join(tmp_path, "lnv")
    hf.save_pretrained(path, safe_serialization=True)
    rng = np.random.default_rng(4)
    video = (rng.random((3, 40, 52, 3)) * 255).astype(np.uint8)
    image = (rng.random((40, 90, 3)) * 255).astype(np.uint8)
    reqs = [{"prompt_token_ids": [5, 17, 301, 99, 3], "multi_modal_data": {"video": video}},
            {"prompt_token_ids": [300, 7, 8, 9], "multi_modal_data": {"image": image}},
            {"prompt_token_ids": [4, 300, 6, 301, 8], "multi_modal_data": {"image": image, "video": video}}]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=8,
              max_num_batched_tokens=128)
    outs = llm.generate(reqs, SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    llm.shutdown()
    proc = FrameVideoProcessor(300, 28, 7, pins, CLIP_MEAN, CLIP_STD, 0, video_token_id=301, pool_stride=2)
    for r, o in zip(reqs, outs):
        ids, inputs, _, ranges = proc(r["prompt_token_ids"], r["multi_modal_data"])
        kw = {}
        for it in inputs:
            if len(it) == 1:
                kw["pixel_values_videos"] = it[0][None]
            else:
                kw["pixel_values"], kw["image_sizes"] = it[0][None], it[1][None]
        assert sum(rg.length for rg in ranges) == sum(t in (300, 301) for t in ids)
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), max_new_tokens=6, do_sample=False, pad_token_id=0,
                              eos_token_id=None, **kw)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, list(r["multi_modal_data"])


def test_bart_matches_hf(tmp_path):
    """Text encoder-decoder: a plain prompt is the encoder input, the decoder starts from its start token and reads
    the encoder states through cross-attention with per-request source lengths; greedy tokens equal transformers'."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.BartConfig(vocab_size=320, d_model=128, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
                        decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256, max_position_embeddings=128,
                        decoder_start_token_id=2, forced_bos_token_id=None, forced_eos_token_id=None, pad_token_id=1,
                        bos_token_id=0, eos_token_id=2, scale_embedding=True)
    hf = tf.BartForConditionalGeneration(cfg).float().eval()
    with torch.no_grad():
        hf.final_logits_bias.normal_(0, 0.3)
    path = os.path.join(tmp_path, "bart")
    hf.save_pretrained(path, safe_serialization=True)
    from safetensors.torch import load_file, save_file
    sd = load_file(os.path.join(path, "model.safetensors"))
    sd["final_logits_bias"] = hf.final_logits_bias.detach().clone()       # a buffer transformers does not persist
    save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    sources = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 13, [4, 6]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=8,
              max_num_batched_tokens=64)
    outs = llm.generate(prompt_token_ids=sources, sampling_params=SamplingParams(temperature=0, max_tokens=8,
                                                                                   ignore_eos=True))
    # explicit decoder prompt
    dec = [2, 40, 41]
    out2 = llm.generate([{"prompt_token_ids": dec, "multi_modal_data": {"encoder_token_ids": sources[1]}}],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))[0]
    llm.shutdown()
    for s_, o in zip(sources, outs):
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([s_]), max_new_tokens=8, do_sample=False, num_beams=1,
                              eos_token_id=None, pad_token_id=1)[0, 1:].tolist()
        assert list(o.outputs[0].token_ids) == ref, s_
    with torch.no_grad():
        ref = hf.generate(input_ids=torch.tensor([sources[1]]), decoder_input_ids=torch.tensor([dec]), max_new_tokens=6,
                          do_sample=False, num_beams=1, eos_token_id=None, pad_token_id=1)[0, len(dec):].tolist()
    assert list(out2.outputs[0].token_ids) == ref


def test_tarsier_split_tokens_over_llava(tmp_path):
    """Tarsier = LLaVA + the text embeddings of a newline token behind every patch row and of an image-end token
    behind the image.  No transformers class exists for it (parity unpinned): the check is structural, against
    this package's own LLaVA features of the same checkpoint, plus an end-to-end run."""
    import json
    import transformers as tf
    torch.manual_seed(0)
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, head_dim=64)
    vc = tf.CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                             image_size=28, patch_size=7, projection_dim=64)
    cfg = tf.LlavaConfig(vision_config=vc, text_config=tc, image_token_index=300, vision_feature_layer=-2)
    hf = tf.LlavaForConditionalGeneration(cfg).float().eval()
    path = os.path.join(tmp_path, "tarsier")
    hf.save_pretrained(path, safe_serialization=True)
    c = json.load(open(os.path.join(path, "config.json")))
    c.update(architectures=["TarsierForConditionalGeneration"], image_newline_idx=301, image_new_idx=302)
    json.dump(c, open(os.path.join(path, "config.json"), "w"))
    img = (np.random.default_rng(5).random((40, 52, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=4)
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    from vllm_distributed_amd.models.llava import LlavaForConditionalGeneration
    from vllm_distributed_amd.multimodal import to_pixel_values
    px = to_pixel_values(img, 28)[None]
    with torch.no_grad():
        t = model.encode_images(_to(model, px)).cpu()
        plain = LlavaForConditionalGeneration.encode_images(model, _to(model, px)).cpu()
        emb = model.language_model.get_input_embeddings(_to(model, torch.tensor([301, 302]))).cpu()
    assert t.shape == (1, 4 * 5 + 1, 128)
    grid = t[0, :20].view(4, 5, 128)
    torch.testing.assert_close(grid[:, :4].reshape(16, 128), plain[0])
    torch.testing.assert_close(grid[:, 4], emb[0].expand(4, 128))
    torch.testing.assert_close(t[0, 20], emb[1])
    out = llm.generate([{"prompt_token_ids": [5, 300, 9], "multi_modal_data": {"image": img}}],
                       SamplingParams(temperature=0, max_tokens=4, ignore_eos=True))[0]
    assert len(out.prompt_token_ids) == 2 + 21 and len(out.outputs[0].token_ids) == 4
    llm.shutdown()


@pytest.mark.parametrize("text", ["gemma", "gemma2"])
def test_paligemma_greedy_matches_hf(tmp_path, text):
    """PaliGemma (Gemma decoder) and PaliGemma 2 (Gemma-2 decoder) against transformers' greedy generation; without
    ``token_type_ids`` transformers runs the prompt causally, which is what the reference's engine does too."""
    import transformers as tf
    torch.manual_seed(0)
    kw = dict(vocab_size=400, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
              num_key_value_heads=1, max_position_embeddings=512, head_dim=64)
    tc = tf.GemmaConfig(**kw) if text == "gemma" else tf.Gemma2Config(**kw, sliding_window=64, query_pre_attn_scalar=64)
    tc.use_bidirectional_attention = False       # transformers' default for this wrapper lets the prompt attend both ways
    vc = tf.SiglipVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                               image_size=28, patch_size=7, projection_dim=128)
    cfg = tf.PaliGemmaConfig(vision_config=vc, text_config=tc, image_token_index=399, projection_dim=128,
                             hidden_size=128, vocab_size=400)
    hf = tf.PaliGemmaForConditionalGeneration(cfg).float().eval()
    path = os.path.join(tmp_path, "pg")
    hf.save_pretrained(path, safe_serialization=True)
    img = (np.random.default_rng(3).random((33, 47, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4)
    out = llm.generate([{"prompt_token_ids": [399, 2, 17, 45, 108], "multi_modal_data": {"image": img}}],
                       SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
    ids = out.prompt_token_ids
    assert ids.count(399) == 16
    from vllm_distributed_amd.multimodal import process_multimodal
    eng = llm.llm_engine
    _, inputs, _, _ = process_multimodal(eng.vllm_config, [399, 2, 17, 45, 108], {"image": img})
    with torch.no_grad():
        ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=inputs[0][None].float(), max_new_tokens=8,
                          do_sample=False, eos_token_id=None, pad_token_id=0)[0, len(ids):].tolist()
    assert list(out.outputs[0].token_ids) == ref
    llm.shutdown()


def test_nvlm_d_dummy_heads_and_projector(tmp_path):
    """NVLM-D = InternVL with 7 all-zero heads beside the tower's own and a bias-free projector as wide as the
    decoder's MLP.  No transformers class exists (parity unpinned): zero dummy heads must leave the tower's output
    unchanged, and a dummy-weight engine must run an image request end to end."""
    import json
    from vllm_distributed_amd.models.internvl import InternVisionLayer
    torch.manual_seed(0)
    a = InternVisionLayer(64, 2, 128, True, False, "layer_norm", 1e-6, "gelu").eval()
    b = InternVisionLayer(64, 2, 128, True, False, "layer_norm", 1e-6, "gelu", dummy_heads=7).eval()
    with torch.no_grad():
        sd = a.state_dict()
        for n, p in b.named_parameters():
            if n == "attn.qkv.weight":
                p.zero_()
                for i in range(3):
                    p[i * 288:i * 288 + 64] = sd[n][i * 64:(i + 1) * 64]
            elif n == "attn.qkv.bias":
                p.zero_()
                for i in range(3):
                    p[i * 288:i * 288 + 64] = sd[n][i * 64:(i + 1) * 64]
            elif n == "attn.proj.weight":
                p.zero_()
                p[:, :64] = sd[n]
            else:
                p.copy_(sd[n])
        x = torch.randn(2, 17, 64)
        torch.testing.assert_close(b(x), a(x))
    path = os.path.join(tmp_path, "nvlm")
    os.makedirs(path)
    cfg = {"architectures": ["NVLM_D"], "model_type": "NVLM_D", "downsample_ratio": 0.5, "select_layer": -1,
           "ps_version": "v2", "dynamic_image_size": True, "max_dynamic_patch": 6, "min_dynamic_patch": 1,
           "use_thumbnail": True, "image_token_id": 300,
           "llm_config": {"architectures": ["Qwen2ForCausalLM"], "model_type": "qwen2", "vocab_size": 320,
                          "hidden_size": 128, "intermediate_size": 256, "num_hidden_layers": 2,
                          "num_attention_heads": 2, "num_key_value_heads": 1, "max_position_embeddings": 512,
                          "rms_norm_eps": 1e-6, "rope_theta": 10000.0, "tie_word_embeddings": False},
           "vision_config": {"hidden_size": 64, "intermediate_size": 128, "num_hidden_layers": 2,
                             "num_attention_heads": 2, "image_size": 28, "patch_size": 7, "qk_normalization": True,
                             "qkv_bias": False, "norm_type": "rms_norm", "layer_norm_eps": 1e-6, "hidden_act": "gelu"}}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=4,
              load_format="dummy")
    m = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    assert m.vision_model.encoder.layers[0].attn.qkv.weight.shape == (3 * 9 * 32, 64)
    assert m.mlp1[1].weight.shape == (256, 256) and m.mlp1[1].bias is None and m.mlp1[3].weight.shape == (128, 256)
    img = (np.random.default_rng(3).random((30, 95, 3)) * 255).astype(np.uint8)
    out = llm.generate([{"prompt_token_ids": [5, 300, 9], "multi_modal_data": {"image": img}}],
                       SamplingParams(temperature=0, max_tokens=4, ignore_eos=True))[0]
    assert len(out.prompt_token_ids) == 2 + 4 * 4 and len(out.outputs[0].token_ids) == 4
    llm.shutdown()


def test_jina_vl_ranking_scores(tmp_path):
    """jina-reranker-m0's form: Qwen2-VL + two-layer scorer on the last hidden state, sigmoid(logit - 2.65).
    Checked against transformers' hidden states of the same Qwen2-VL weights and the scorer applied by hand."""
    import json
    from safetensors.torch import load_file, save_file
    path, hf = _tiny_qwen2_vl(str(tmp_path))
    torch.manual_seed(4)
    H = hf.config.get_text_config().hidden_size
    score = torch.nn.Sequential(torch.nn.Linear(H, H), torch.nn.ReLU(), torch.nn.Linear(H, 1)).float()
    f = [x for x in os.listdir(path) if x.endswith(".safetensors")][0]
    sd = load_file(os.path.join(path, f))
    sd.update({"score." + n: t.detach().clone() for n, t in score.state_dict().items()})
    save_file(sd, os.path.join(path, f))
    c = json.load(open(os.path.join(path, "config.json")))
    c.update(architectures=["JinaVLForRanking"], num_labels=1)
    json.dump(c, open(os.path.join(path, "config.json"), "w"))
    g = torch.Generator().manual_seed(3)
    img = {"pixel_values": torch.randn(4 * 6, 1176, generator=g), "image_grid_thw": torch.tensor([1, 4, 6])}
    prompts = [([5, 17, 302, 300, 303, 9, 11, 100], [img]), ([7, 8, 9, 10, 100], [])]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              enable_prefix_caching=False)
    outs = llm.classify([{"prompt_token_ids": p, **({"multi_modal_data": {"image": m}} if m else {})}
                         for p, m in prompts])
    for (p, ims), o in zip(prompts, outs):
        ids = []
        for t in p:
            ids += [t] * (int(ims[0]["image_grid_thw"].prod()) // 4 if t == 300 else 1)
        kw = {}
        if ims:
            kw = dict(pixel_values=ims[0]["pixel_values"], image_grid_thw=ims[0]["image_grid_thw"][None],
                      mm_token_type_ids=torch.tensor([[1 if t == 300 else 0 for t in ids]], dtype=torch.int32))
        with torch.no_grad():
            h = hf(input_ids=torch.tensor([ids]), output_hidden_states=True, **kw).hidden_states[-1][0, -1]
            ref = torch.sigmoid(score(h) - 2.65)
        got = torch.as_tensor(o.outputs.probs if hasattr(o.outputs, "probs") else o.outputs.data).float().view(-1)
        torch.testing.assert_close(got, ref.view(-1), atol=2e-4, rtol=1e-3)
    llm.shutdown()


def test_aya_vision_matches_hf(tmp_path):
    """SigLIP tower -> pixel shuffle -> LayerNorm + SwiGLU projector over a Cohere-2 decoder, one image cut into
    tiles + thumbnail and one single-tile image: greedy tokens equal transformers'."""
    import transformers as tf
    from vllm_distributed_amd.multimodal import process_multimodal
    torch.manual_seed(0)
    tc = tf.Cohere2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                          num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, sliding_window=64,
                          pad_token_id=0, bos_token_id=1, eos_token_id=2)
    vc = tf.SiglipVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                               image_size=28, patch_size=7)
    cfg = tf.AyaVisionConfig(vision_config=vc, text_config=tc, image_token_index=300, downsample_factor=2,
                             vision_feature_select_strategy="full", vision_feature_layer=-1)
    hf = tf.AyaVisionForConditionalGeneration(cfg).float().eval()
    path = os.path.join(tmp_path, "aya")
    hf.save_pretrained(path, safe_serialization=True)
    rng = np.random.default_rng(3)
    imgs = [(rng.random(s + (3,)) * 255).astype(np.uint8) for s in ((30, 95), (28, 28))]
    prompts = [[5, 17, 300, 99, 3], [4, 300, 6]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=8)
    reqs = [{"prompt_token_ids": p, "multi_modal_data": {"image": im}} for p, im in zip(prompts, imgs)]
    outs = llm.generate(reqs, SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    cfg_v = llm.llm_engine.vllm_config
    for p, im, o in zip(prompts, imgs, outs):
        ids, inputs, _, _ = process_multimodal(cfg_v, p, {"image": im})
        tiles = inputs[0][0]
        assert len(ids) == len(p) - 1 + tiles.shape[0] * 4
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=tiles, max_new_tokens=6, do_sample=False,
                              pad_token_id=0, eos_token_id=None)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, im.shape
    llm.shutdown()


@pytest.mark.parametrize("name", ["Idefics3", "SmolVLM"])
def test_idefics3_smolvlm_match_hf(tmp_path, name):
    """SigLIP-style tower with its closing norm, pixel shuffle, one projection, Llama decoder; a wide image (tiles +
    whole image) given as one placeholder, and the same image given the way a chat template expands it (one run of
    image tokens per tile with text between): greedy tokens equal transformers'."""
    import transformers as tf
    from vllm_distributed_amd.multimodal import process_multimodal
    torch.manual_seed(0)
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, head_dim=64,
                        pad_token_id=0)
    Cfg, VCfg, Model = ((tf.Idefics3Config, tf.Idefics3VisionConfig, tf.Idefics3ForConditionalGeneration)
                        if name == "Idefics3" else
                        (tf.SmolVLMConfig, tf.SmolVLMVisionConfig, tf.SmolVLMForConditionalGeneration))
    vc = VCfg(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=28,
              patch_size=7)
    cfg = Cfg(vision_config=vc, text_config=tc, image_token_id=300, scale_factor=2, pad_token_id=0)
    hf = Model(cfg).float().eval()
    path = os.path.join(tmp_path, name)
    hf.save_pretrained(path, safe_serialization=True)
    img = (np.random.default_rng(3).random((24, 80, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=8,
              hf_overrides={"longest_edge": 84})
    cfg_v = llm.llm_engine.vllm_config
    ids1, inputs, _, _ = process_multimodal(cfg_v, [5, 17, 300, 99, 3], {"image": img})
    tiles = inputs[0][0]
    assert tiles.shape[0] == 3 + 1 and len(ids1) == 4 + 4 * 4           # 1 x 3 grid + the whole image
    expanded = [5]
    for j in range(4):
        expanded += [310 + j] + [300] * 4                                 # a text tag in front of every tile
    expanded += [99, 3]
    reqs = [{"prompt_token_ids": [5, 17, 300, 99, 3], "multi_modal_data": {"image": img}},
            {"prompt_token_ids": expanded, "multi_modal_data": {"image": img}}]
    outs = llm.generate(reqs, SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    for ids, o in zip((ids1, expanded), outs):
        assert list(o.prompt_token_ids) == ids
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=tiles[None],
                              pixel_attention_mask=torch.ones(1, 4, 28, 28, dtype=torch.bool), max_new_tokens=6,
                              do_sample=False, pad_token_id=0, eos_token_id=None)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref
    llm.shutdown()


@pytest.mark.parametrize("ln_mid", [False, True])
def test_ultravox_matches_hf_components(tmp_path, ln_mid):
    """Ultravox has no transformers class (remote code upstream; parity unpinned as a whole).  The reference here is
    assembled from transformers' own parts: its Whisper encoder on the clip's valid frames, the frame stacking and
    projector written out, and its Llama decoder run from input embeddings."""
    import json
    import transformers as tf
    from safetensors.torch import save_file
    from vllm_distributed_amd.multimodal import process_multimodal
    torch.manual_seed(0)
    wc = tf.WhisperConfig(num_mel_bins=80, d_model=64, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=128,
                          max_source_positions=1500, decoder_layers=1, decoder_attention_heads=2, decoder_ffn_dim=64,
                          vocab_size=100, pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1)
    enc = tf.WhisperModel(wc).encoder.float().eval()
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, head_dim=64,
                        tie_word_embeddings=False)
    lm = tf.LlamaForCausalLM(tc).float().eval()
    stack, hid = 8, 96
    g = lambda *s: torch.randn(*s) * 0.2
    mid = hid // 2
    proj = {"ln_pre.weight": 1 + g(64 * stack) * 0.1, "linear_1.weight": g(hid, 64 * stack), "linear_2.weight": g(128, mid),
            ("ln_mid.weight" if ln_mid else "ln_post.weight"): 1 + g(mid if ln_mid else 128) * 0.1}
    sd = {"audio_tower." + n: t for n, t in enc.state_dict().items()}
    sd.update({"multi_modal_projector." + n: t for n, t in proj.items()})
    sd.update({"language_model." + n: t for n, t in lm.state_dict().items()})
    path = os.path.join(tmp_path, "uv")
    os.makedirs(path)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    json.dump({"architectures": ["UltravoxModel"], "model_type": "ultravox", "audio_config": wc.to_dict(),
               "text_config": tc.to_dict(), "stack_factor": stack, "hidden_size": hid, "projector_act": "swiglu",
               "projector_ln_mid": ln_mid, "audio_token_index": 300, "torch_dtype": "float32"},
              open(os.path.join(path, "config.json"), "w"))
    rng = np.random.default_rng(3)
    clip = rng.standard_normal(16000 * 2).astype(np.float32) * 0.1
    prompt = [5, 17, 300, 99, 3]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=4)
    out = llm.generate([{"prompt_token_ids": prompt, "multi_modal_data": {"audio": clip}}],
                       SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
    ids, inputs, _, _ = process_multimodal(llm.llm_engine.vllm_config, prompt, {"audio": clip})
    llm.shutdown()
    feats, mel = inputs[0][0], int(inputs[0][1])
    frames = (mel - 1) // 2 + 1
    ntok = -(-frames // stack)
    assert list(out.prompt_token_ids) == ids and ids.count(300) == ntok == 13
    F = torch.nn.functional
    rms = lambda x, w: x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    with torch.no_grad():
        # the encoder on the valid frames only equals the padded run with the padding masked out
        x = F.gelu(enc.conv1(feats[None])); x = F.gelu(enc.conv2(x)).permute(0, 2, 1)
        x = x + enc.embed_positions.weight[: x.shape[1]]
        mask = torch.zeros(1, 1, 1, x.shape[1]); mask[..., frames:] = torch.finfo(torch.float32).min
        for layer in enc.layers:
            x = layer(x, attention_mask=mask)
            x = x[0] if isinstance(x, tuple) else x
        x = enc.layer_norm(x)
        pad = -x.shape[1] % stack
        x = F.pad(x, (0, 0, 0, pad)).reshape(1, -1, 64 * stack)
        hh = rms(x, proj["ln_pre.weight"]) @ proj["linear_1.weight"].T
        a, b = hh.chunk(2, -1)
        hh = a * F.silu(b)
        if ln_mid:
            hh = rms(hh, proj["ln_mid.weight"])
        hh = hh @ proj["linear_2.weight"].T
        if not ln_mid:
            hh = rms(hh, proj["ln_post.weight"])
        emb = lm.model.embed_tokens(torch.tensor([ids]))
        emb[0, torch.tensor(ids) == 300] = hh[0, :ntok]
        ref = lm.generate(inputs_embeds=emb, max_new_tokens=8, do_sample=False, pad_token_id=0,
                          eos_token_id=None)[0].tolist()
    assert list(out.outputs[0].token_ids) == ref[-8:]


def test_minimax_vl_01_runs_llava_next_path_over_minimax_decoder(tmp_path):
    """MiniMax-VL-01 = LLaVA-NeXT's image path over the MiniMax decoder.  No transformers class (parity unpinned);
    both halves have their own transformers parity tests, this one checks the composition: a LLaVA-NeXT checkpoint
    whose text model is transformers' MiniMax, loaded under the MiniMax-VL-01 name, against that same composition
    run by transformers' LLaVA-NeXT class."""
    import json
    import transformers as tf
    from vllm_distributed_amd.multimodal import process_multimodal
    torch.manual_seed(0)
    tc = tf.MiniMaxConfig(vocab_size=320, hidden_size=128, intermediate_size=192, num_hidden_layers=3,
                          num_attention_heads=2, num_key_value_heads=1, head_dim=64, num_local_experts=4,
                          num_experts_per_tok=2, max_position_embeddings=512, block_size=16,
                          layer_types=["linear_attention", "full_attention", "linear_attention"])
    vc = tf.CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                             image_size=28, patch_size=7, projection_dim=64)
    cfg = tf.LlavaNextConfig(vision_config=vc, text_config=tc, image_token_index=300, vision_feature_layer=-2,
                             image_grid_pinpoints=[[28, 28], [28, 56], [56, 28]])
    hf = tf.LlavaNextForConditionalGeneration(cfg).float().eval()
    path = os.path.join(tmp_path, "mmvl")
    hf.save_pretrained(path, safe_serialization=True)
    c = json.load(open(os.path.join(path, "config.json")))
    c["architectures"] = ["MiniMaxVL01ForConditionalGeneration"]
    json.dump(c, open(os.path.join(path, "config.json"), "w"))
    img = (np.random.default_rng(5).random((30, 52, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=4)
    prompt = [5, 17, 300, 99, 3]
    out = llm.generate([{"prompt_token_ids": prompt, "multi_modal_data": {"image": img}}],
                       SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))[0]
    ids, inputs, _, _ = process_multimodal(llm.llm_engine.vllm_config, prompt, {"image": img})
    llm.shutdown()
    with torch.no_grad():
        cur, ref = list(ids), []
        for _ in range(6):               # no cache: transformers' LLaVA-NeXT wrapper cannot build MiniMax's own
            lg = hf(input_ids=torch.tensor([cur]), pixel_values=inputs[0][0][None],
                    image_sizes=torch.tensor([[int(x) for x in inputs[0][1]]]), use_cache=False).logits[0, -1]
            ref.append(int(lg.argmax()))
            cur.append(ref[-1])
    assert list(out.outputs[0].token_ids) == ref


@pytest.mark.parametrize("swin", [False, True])
def test_chameleon_greedy_matches_hf(tmp_path, swin):
    """Chameleon: VQ-VAE image codes -> vocabulary ids -> the decoder's own embedding table; per-head LayerNorm on
    q / k; both norm placements.  Against transformers' greedy generation on the expanded prompt."""
    import transformers as tf
    torch.manual_seed(0)
    vocab = {"<image>": 300, "<racm3:break>": 301, "<eoss>": 302}
    for i in range(64):                                   # 64 codes, spelled with A..J digits
        vocab["IMGIMG" + "".join(chr(ord("A") + int(c)) for c in str(i)) + "Z"] = 200 + (i * 37) % 64
    vq = dict(embed_dim=16, num_embeddings=64, latent_channels=16, resolution=32, in_channels=3, base_channels=32,
              channel_multiplier=[1, 2, 2], num_res_blocks=1, attn_resolutions=[16], attn_type="vanilla")
    cfg = tf.ChameleonConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                             num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                             swin_norm=swin, vq_config=vq, vocabulary_map=vocab, image_token_id=300)
    hf = tf.ChameleonForConditionalGeneration(cfg).float().eval()
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if "q_norm" in n or "k_norm" in n:
                p.uniform_(0.5, 1.5) if n.endswith("weight") else p.normal_(0, 0.2)
    path = os.path.join(tmp_path, "cham")
    hf.save_pretrained(path, safe_serialization=True)
    px = torch.randn(3, 32, 32, generator=torch.Generator().manual_seed(2))
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4)
    out = llm.generate([{"prompt_token_ids": [5, 300, 17, 45], "multi_modal_data": {"image": px}}],
                       SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
    ids = out.prompt_token_ids
    assert ids[:2] == [5, 301] and ids.count(300) == 64 and ids[66] == 302
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    with torch.no_grad():
        codes = model.image_tokens(_to(model, px[None])).cpu()
        ref_codes = hf.model.get_image_tokens(px[None])
        assert torch.equal(codes, ref_codes.to(codes.dtype))
        ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=px[None], max_new_tokens=8, do_sample=False,
                          eos_token_id=None, pad_token_id=0)[0, len(ids):].tolist()
    got = list(out.outputs[0].token_ids)
    llm.shutdown()
    assert got == ref
    assert not any(200 <= t < 264 for t in got)           # image codes are masked out of the logits


def test_blip2_greedy_matches_hf(tmp_path):
    """BLIP-2 (ViT + Q-Former + projection over OPT) against transformers' greedy generation."""
    import transformers as tf
    torch.manual_seed(0)
    vc = tf.Blip2VisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                              image_size=28, patch_size=7)
    qc = tf.Blip2QFormerConfig(vocab_size=100, hidden_size=48, num_hidden_layers=3, num_attention_heads=2,
                               intermediate_size=96, encoder_hidden_size=64, cross_attention_frequency=2)
    tc = tf.OPTConfig(vocab_size=320, hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=2,
                      max_position_embeddings=256, word_embed_proj_dim=128)
    cfg = tf.Blip2Config(vision_config=vc.to_dict(), qformer_config=qc.to_dict(), text_config=tc.to_dict(),
                         num_query_tokens=8, image_token_index=300)
    hf = tf.Blip2ForConditionalGeneration(cfg).float().eval()
    with torch.no_grad():
        hf.query_tokens.normal_(0, 1.0)
        for n, p in hf.named_parameters():
            if p.dim() >= 2 and "language_model" not in n:
                p.mul_(2.0)
    path = os.path.join(tmp_path, "blip2")
    hf.save_pretrained(path, safe_serialization=True)
    img = (np.random.default_rng(3).random((33, 47, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4)
    out = llm.generate([{"prompt_token_ids": [2, 300, 17, 45, 108], "multi_modal_data": {"image": img}}],
                       SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
    ids = out.prompt_token_ids
    assert ids.count(300) == 8
    from vllm_distributed_amd.multimodal import process_multimodal
    _, inputs, _, _ = process_multimodal(llm.llm_engine.vllm_config, [2, 300, 17, 45, 108], {"image": img})
    with torch.no_grad():
        ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=inputs[0][None].float(), max_new_tokens=8,
                          do_sample=False, eos_token_id=None, pad_token_id=1)[0].tolist()
    ref = ref[-8:]
    llm.shutdown()
    assert list(out.outputs[0].token_ids) == ref


def test_glm4v_matches_hf(tmp_path):
    """GLM-4.1V: resampled position table, full-frame vision attention, convolutional 2 x 2 merge, gated merger,
    GLM-4 decoder with interleaved partial multimodal rotary positions; two images with different grids."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.Glm4vConfig(
        text_config=dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                         rope_parameters={"rope_type": "default", "mrope_section": [4, 6, 6], "rope_theta": 10000.0,
                                          "partial_rotary_factor": 0.5}, pad_token_id=0),
        vision_config=dict(depth=2, hidden_size=64, intermediate_size=96, num_heads=2, out_hidden_size=128,
                           patch_size=14, spatial_merge_size=2, temporal_patch_size=2, image_size=112, in_channels=3),
        image_token_id=300, video_token_id=301, image_start_token_id=302, image_end_token_id=303)
    hf = tf.Glm4vForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)
    hf = hf.float().eval()
    path = os.path.join(str(tmp_path), "glm4v")
    hf.save_pretrained(path, safe_serialization=True)
    g = torch.Generator().manual_seed(5)
    imgs = [{"pixel_values": torch.randn(6 * 10, 1176, generator=g), "image_grid_thw": torch.tensor([1, 6, 10])},
            {"pixel_values": torch.randn(4 * 4, 1176, generator=g), "image_grid_thw": torch.tensor([1, 4, 4])}]
    p = [5, 17, 302, 300, 303, 9, 302, 300, 303, 11, 250]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64, enable_prefix_caching=False, block_size=16)
    o = llm.generate([{"prompt_token_ids": p, "multi_modal_data": {"image": imgs}}],
                     SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
    ids = [5, 17, 302] + [300] * 15 + [303, 9, 302] + [300] * 4 + [303, 11, 250]
    assert list(o.prompt_token_ids) == ids
    with torch.no_grad():
        ref = hf.generate(input_ids=torch.tensor([ids]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                          eos_token_id=None, pixel_values=torch.cat([i["pixel_values"] for i in imgs]),
                          image_grid_thw=torch.stack([i["image_grid_thw"] for i in imgs]),
                          mm_token_type_ids=torch.tensor([[1 if t == 300 else 0 for t in ids]], dtype=torch.int32)
                          )[0, len(ids):].tolist()
    assert list(o.outputs[0].token_ids) == ref
    llm.shutdown()


def test_aria_matches_hf(tmp_path):
    """Aria: tower over the valid rectangle of a padded square (bucketed position ids), query projector whose
    query count follows the padded size, expert decoder with shared experts; one small padded image and one large."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.AriaConfig(
        vision_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                           image_size=56, patch_size=14, model_type="idefics3_vision"),
        text_config=dict(vocab_size=320, hidden_size=128, intermediate_size=64, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=512,
                         moe_num_experts=4, moe_topk=2, moe_num_shared_experts=2, pad_token_id=0,
                         model_type="aria_text"),
        projector_patch_to_query_dict={16: 4, 64: 8}, image_token_index=300, vision_feature_layer=-1)
    hf = tf.AriaForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0) if "experts.fc" not in n else p.normal_(0, 0.08)
        hf.model.multi_modal_projector.query.normal_(0, 1.0)
    hf = hf.float().eval()
    path = os.path.join(str(tmp_path), "aria")
    hf.save_pretrained(path, safe_serialization=True)
    g = torch.Generator().manual_seed(5)
    small = torch.zeros(3, 56, 56)
    small[:, :56, :40] = torch.randn(3, 56, 40, generator=g)             # 4 x 3 valid patches (the last one partly)
    large = torch.zeros(3, 112, 112)
    large[:, :70, :112] = torch.randn(3, 70, 112, generator=g)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64, enable_prefix_caching=False, block_size=16)
    for px, valid, nq in ((small, (56, 40), 4), (large, (70, 112), 8)):
        p = [5, 17, 300, 9, 11, 250]
        o = llm.generate([{"prompt_token_ids": p,
                           "multi_modal_data": {"image": {"pixel_values": px, "valid_size": valid}}}],
                         SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
        ids = [5, 17] + [300] * nq + [9, 11, 250]
        assert list(o.prompt_token_ids) == ids
        mask = torch.zeros(1, px.shape[-2], px.shape[-1], dtype=torch.bool)
        mask[:, :valid[0], :valid[1]] = True
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), pixel_values=px[None], pixel_mask=mask, max_new_tokens=8,
                              do_sample=False, pad_token_id=0, eos_token_id=None)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, valid
    llm.shutdown()


def test_granite_speech_matches_hf(tmp_path):
    """Granite Speech: conformer encoder (block attention with a padded last block, relative positions, batch-norm
    statistics, CTC feedback), windowed Q-Former projector, Granite decoder; two clips of different lengths.
    The mel front end is checked for shape only (transformers' needs torchaudio, which is not installed)."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.GraniteSpeechConfig(
        encoder_config=dict(input_dim=160, num_layers=4, hidden_dim=64, feedforward_mult=2, num_heads=2, dim_head=32,
                            output_dim=12, context_size=10, max_pos_emb=16, conv_kernel_size=5,
                            conv_expansion_factor=2),
        projector_config=dict(model_type="blip_2_qformer", hidden_size=48, num_hidden_layers=2, num_attention_heads=2,
                              intermediate_size=96, encoder_hidden_size=64, cross_attention_frequency=1,
                              vocab_size=10),
        text_config=dict(model_type="granite", vocab_size=320, hidden_size=128, intermediate_size=256,
                         num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                         max_position_embeddings=512, embedding_multiplier=3.0, residual_multiplier=0.5,
                         attention_multiplier=0.2, logits_scaling=2.0, pad_token_id=0, tie_word_embeddings=True),
        audio_token_index=300, downsample_rate=3, window_size=6, has_lora_adapter=False)
    hf = tf.GraniteSpeechForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)
        for m in hf.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
    hf = hf.float().eval()
    path = os.path.join(str(tmp_path), "gs")
    hf.save_pretrained(path, safe_serialization=True)
    g = torch.Generator().manual_seed(5)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64, enable_prefix_caching=False, block_size=16)
    for frames in (23, 30):
        feats = torch.randn(frames, 160, generator=g)
        p = [5, 17, 300, 9, 11, 250]
        o = llm.generate([{"prompt_token_ids": p, "multi_modal_data": {"audio": {"input_features": feats}}}],
                         SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
        ntok = -(-frames // 6) * 2
        ids = [5, 17] + [300] * ntok + [9, 11, 250]
        assert list(o.prompt_token_ids) == ids
        with torch.no_grad():
            ref = hf.generate(input_ids=torch.tensor([ids]), input_features=feats[None],
                              input_features_mask=torch.ones(1, ntok, dtype=torch.bool), max_new_tokens=8,
                              do_sample=False, pad_token_id=0, eos_token_id=None)[0, len(ids):].tolist()
        assert list(o.outputs[0].token_ids) == ref, frames
    # waveform front end: one second of audio -> 101 mel frames -> 50 stacked frames -> ceil(50 / 6) * 2 tokens
    from vllm_distributed_amd.multimodal import process_multimodal
    wave = torch.randn(16000, generator=g).numpy()
    ids, inputs, _, pos = process_multimodal(llm.llm_engine.vllm_config, [5, 300, 9], {"audio": wave})
    assert tuple(inputs[0][0].shape) == (50, 160) and pos[0].length == 18 and ids.count(300) == 18
    llm.shutdown()


def test_qwen2_5_omni_thinker_matches_hf(tmp_path):
    """Qwen2.5-Omni thinker: windowed audio encoder (a clip of two full windows and a short one), the vision tower
    with split q / k / v weights, audio and image in one prompt with multimodal rotary positions."""
    from transformers.models.qwen2_5_omni import configuration_qwen2_5_omni as c, modeling_qwen2_5_omni as m
    torch.manual_seed(0)
    cfg = c.Qwen2_5OmniThinkerConfig(
        audio_config=dict(num_mel_bins=16, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=64, d_model=32,
                          output_dim=128, n_window=4, max_source_positions=50),
        vision_config=dict(depth=2, hidden_size=64, intermediate_size=96, num_heads=2, out_hidden_size=128,
                           patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=56,
                           fullatt_block_indexes=[1], in_channels=3),
        text_config=dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                         rope_parameters={"rope_type": "default", "mrope_section": [8, 12, 12],
                                          "rope_theta": 10000.0}),
        audio_token_id=304, image_token_id=300, video_token_id=301, audio_start_token_id=305, audio_end_token_id=306,
        vision_start_token_id=302, vision_end_token_id=303)
    hf = m.Qwen2_5OmniThinkerForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)
    hf = hf.float().eval()
    path = os.path.join(str(tmp_path), "omni")
    hf.save_pretrained(path, safe_serialization=True)
    import json
    cj = json.load(open(os.path.join(path, "config.json")))
    cj["architectures"] = ["Qwen2_5OmniModel"]
    json.dump(cj, open(os.path.join(path, "config.json"), "w"))
    g = torch.Generator().manual_seed(5)
    mel = torch.randn(16, 21, generator=g)                          # windows of 8 mel frames: 8 + 8 + 5
    img = {"pixel_values": torch.randn(6 * 10, 1176, generator=g), "image_grid_thw": torch.tensor([1, 6, 10])}
    p = [5, 305, 304, 306, 17, 302, 300, 303, 9, 11, 250]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64, enable_prefix_caching=False, block_size=16)
    o = llm.generate([{"prompt_token_ids": p, "multi_modal_data": {"audio": {"input_features": mel}, "image": [img]}}],
                     SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0]
    na = ((21 - 1) // 2 + 1 - 2) // 2 + 1
    ids = [5, 305] + [304] * na + [306, 17, 302] + [300] * 15 + [303, 9, 11, 250]
    assert list(o.prompt_token_ids) == ids
    with torch.no_grad():
        ref = hf.generate(input_ids=torch.tensor([ids]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                          eos_token_id=None, pixel_values=img["pixel_values"], image_grid_thw=img["image_grid_thw"][None],
                          input_features=mel[None], feature_attention_mask=torch.ones(1, 21, dtype=torch.long)
                          )[0, len(ids):].tolist()
    assert list(o.outputs[0].token_ids) == ref
    llm.shutdown()


def test_qwen_vl_tower_and_engine(tmp_path):
    """Qwen-VL (first generation).  No transformers class exists (parity unpinned): the 2-D sine / cosine table is
    checked against the formula written out with numpy, a tower block against attention computed head by head
    from the head-major fused projection, the resampler against the same steps spelled out with
    ``F.multi_head_attention_forward``; then a checkpoint written from the module's own weights runs end to end."""
    import json
    from safetensors.torch import save_file
    from vllm_distributed_amd.models.qwen_vl import QueryResampler, QwenVisionTransformer, _Block, sincos_2d
    torch.manual_seed(0)
    # position table
    dim, h, w = 16, 3, 5
    omega = 1.0 / 10000 ** (np.arange(dim // 4, dtype=np.float32) / (dim / 4.0))
    ref = np.zeros((h * w, dim), dtype=np.float32)
    for r in range(h):
        for c in range(w):
            ref[r * w + c] = np.concatenate([np.sin(c * omega), np.cos(c * omega), np.sin(r * omega), np.cos(r * omega)])
    torch.testing.assert_close(sincos_2d(dim, h, w), torch.from_numpy(ref), rtol=1e-5, atol=1e-6)
    # one block
    blk = _Block(32, 2, 2.0).eval()
    x = torch.randn(2, 7, 32)
    with torch.no_grad():
        y = blk.ln_1(x)
        p = blk.attn.in_proj(y).view(2, 7, 2, 48)
        heads = []
        for hd in range(2):
            q, k, v = p[:, :, hd, :16], p[:, :, hd, 16:32], p[:, :, hd, 32:]
            heads.append(torch.softmax(q @ k.transpose(1, 2) / 4.0, -1) @ v)
        z = x + blk.attn.out_proj(torch.cat(heads, -1))
        z = z + blk.mlp.c_proj(torch.nn.functional.gelu(blk.mlp.c_fc(blk.ln_2(z))))
        torch.testing.assert_close(blk(x), z, rtol=1e-4, atol=1e-5)
    # resampler
    rs = QueryResampler(2, 128, 1, 32, post_projection=False).eval()
    with torch.no_grad():
        rs.query.normal_()
        feats = torch.randn(1, 9, 32)
        kv = rs.ln_kv(rs.kv_proj(feats))[0]
        from vllm_distributed_amd.models.qwen_vl import resize_positions
        pos = resize_positions(rs.pos_embed, 3, 3)
        q = rs.ln_q(rs.query) + rs.pos_embed
        wq, wk, wv = rs.attn.in_proj_weight.chunk(3)
        bq, bk, bv = rs.attn.in_proj_bias.chunk(3)
        a = torch.softmax((q @ wq.t() + bq) @ ((kv + pos) @ wk.t() + bk).t() / 128 ** 0.5, -1) @ (kv @ wv.t() + bv)
        torch.testing.assert_close(rs(feats)[0], rs.attn.out_proj(a), rtol=1e-4, atol=1e-5)
    # end to end
    vis = dict(image_size=56, patch_size=14, width=32, layers=2, heads=2, mlp_ratio=2.0, n_queries=4, output_dim=128,
               image_start_id=300)
    cfg = {"architectures": ["QwenVLForConditionalGeneration"], "model_type": "qwen", "vocab_size": 320,
           "hidden_size": 128, "intermediate_size": 512, "num_hidden_layers": 2, "num_attention_heads": 2,
           "kv_channels": 64, "layer_norm_epsilon": 1e-6, "rotary_emb_base": 10000, "seq_length": 512,
           "tie_word_embeddings": False, "visual": vis}
    path = os.path.join(str(tmp_path), "qwenvl")
    os.makedirs(path)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    tower = QwenVisionTransformer(**vis)
    sd = {"transformer.visual." + n: (torch.randn_like(t) * 0.2) for n, t in tower.state_dict().items()}
    sd["transformer.wte.weight"] = torch.randn(320, 128) * 0.5
    sd["transformer.ln_f.weight"] = torch.ones(128)
    sd["lm_head.weight"] = torch.randn(320, 128) * 0.5
    for i in range(2):
        b = f"transformer.h.{i}."
        sd.update({b + "ln_1.weight": torch.ones(128), b + "ln_2.weight": torch.ones(128),
                   b + "attn.c_attn.weight": torch.randn(384, 128) * 0.1, b + "attn.c_attn.bias": torch.randn(384) * 0.1,
                   b + "attn.c_proj.weight": torch.randn(128, 128) * 0.1, b + "mlp.w1.weight": torch.randn(256, 128) * 0.1,
                   b + "mlp.w2.weight": torch.randn(256, 128) * 0.1, b + "mlp.c_proj.weight": torch.randn(128, 256) * 0.1})
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    img = (np.random.default_rng(3).random((40, 52, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4)
    out = llm.generate([{"prompt_token_ids": [5, 302, 9, 11], "multi_modal_data": {"image": img}},
                        {"prompt_token_ids": [5, 9, 11]}],
                       SamplingParams(temperature=0, max_tokens=4, ignore_eos=True))
    assert list(out[0].prompt_token_ids) == [5, 300, 302, 302, 302, 302, 301, 9, 11]
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    tower.load_state_dict({n[len("transformer.visual."):]: t for n, t in sd.items() if ".visual." in n})
    from vllm_distributed_amd.multimodal import process_multimodal
    _, inputs, _, _ = process_multimodal(llm.llm_engine.vllm_config, [5, 302, 9, 11], {"image": img})
    with torch.no_grad():
        torch.testing.assert_close(model.encode_images(_to(model, inputs[0][None])).cpu(), tower(inputs[0][None]),
                                   atol=1e-4, rtol=1e-4)
    assert len(out[0].outputs[0].token_ids) == 4 and out[0].outputs[0].token_ids != out[1].outputs[0].token_ids
    llm.shutdown()


def test_phi3v_hd_stitching_and_engine(tmp_path):
    """Phi-3-vision.  No transformers class exists (parity unpinned).  The high-resolution size rule is checked on
    known cases, the 2 x 2 merge + crop stitching + row separators against a loop over crops, rows and columns, and a
    tiny checkpoint (Phi-3 decoder from transformers, random tower) runs an image request end to end; without images
    the tokens equal transformers' Phi-3."""
    import json
    import transformers as tf
    from safetensors.torch import load_file, save_file
    from vllm_distributed_amd.models.phi3v import Phi3HDImageEmbedding, hd_size
    from vllm_distributed_amd.config import HFConfig
    assert hd_size(1344, 1344) == (1344, 1344, 1344, 1344)                  # 4 x 4 crops
    assert hd_size(2000, 500) == (2688, 672, 2688, 672)                     # 8 x 2 crops
    assert hd_size(500, 2000) == (672, 2688, 672, 2688)
    assert hd_size(800, 600)[2:] == (1344, 1008)                            # 4 x 3 crops, height padded to 3 crops
    torch.manual_seed(0)
    ip = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2, image_size=56,
              patch_size=14, image_dim_out=32)
    emb = Phi3HDImageEmbedding(HFConfig(dict(img_processor=ip, hidden_size=128))).eval()
    with torch.no_grad():
        for p in emb.parameters():
            p.normal_(0, 0.3)
        crops = torch.randn(1 + 6, 3, 56, 56)
        got = emb(crops, (112, 168))                                        # 2 rows x 3 columns of crops
        feats = emb.img_processor.vision_model(crops)[:, 1:]               # [7, 16, 32], a 4 x 4 grid per crop

        def merged(f):                                                      # one crop -> [2, 2, 128]
            f = f.view(4, 4, 32)
            return torch.stack([torch.stack([torch.cat([f[2 * i, 2 * j], f[2 * i, 2 * j + 1], f[2 * i + 1, 2 * j],
                                                        f[2 * i + 1, 2 * j + 1]]) for j in range(2)]) for i in range(2)])
        rows = []
        for r in range(2):
            for i in range(2):
                for c in range(3):
                    rows.extend(merged(feats[1 + r * 3 + c])[i])
                rows.append(emb.sub_GN.view(-1))
        rows.append(emb.glb_GN.view(-1))
        ov = merged(feats[0])
        for i in range(2):
            rows.extend(ov[i])
            rows.append(emb.sub_GN.view(-1))
        torch.testing.assert_close(got, emb.img_projection(torch.stack(rows)))
    assert got.shape[0] == (6 + 1) * 4 + 1 + (2 + 1) * 2
    # end to end
    tc = tf.Phi3Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, pad_token_id=0,
                       bos_token_id=1, eos_token_id=2)
    hf = tf.Phi3ForCausalLM(tc).float().eval()
    path = os.path.join(str(tmp_path), "phi3v")
    hf.save_pretrained(path, safe_serialization=True)
    sd = load_file(os.path.join(path, "model.safetensors"))
    sd.update({"model.vision_embed_tokens." + n: t.detach().clone() for n, t in emb.state_dict().items()})
    save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    cj = json.load(open(os.path.join(path, "config.json")))
    cj.update(architectures=["Phi3VForCausalLM"], img_processor=ip, image_token_id=300)
    json.dump(cj, open(os.path.join(path, "config.json"), "w"))
    img = (np.random.default_rng(3).random((60, 150, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=4,
              mm_processor_kwargs={"num_crops": 4})
    outs = llm.generate([{"prompt_token_ids": [5, 300, 9, 11], "multi_modal_data": {"image": img}},
                         {"prompt_token_ids": [5, 17, 9, 11]}],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    w, h, pw, ph = hd_size(150, 60, 56, 4)
    n = ((ph // 56) * (pw // 56) + 1) * 4 + 1 + (ph // 56 + 1) * 2
    assert list(outs[0].prompt_token_ids).count(300) == n and (pw // 56, ph // 56) == (2, 1)
    with torch.no_grad():
        ref = hf.generate(torch.tensor([[5, 17, 9, 11]]), max_new_tokens=6, do_sample=False, pad_token_id=0,
                          eos_token_id=None)[0, 4:].tolist()
    assert list(outs[1].outputs[0].token_ids) == ref
    assert len(outs[0].outputs[0].token_ids) == 6
    llm.shutdown()


def test_deepseek_vl2_stitching_and_engine(tmp_path):
    """DeepSeek-VL2.  No transformers class exists (parity unpinned).  The 2 x 2 concatenation of the projector and
    the overview / tile-map layout with newline and separator vectors are checked against loops; the timm-named
    SigLIP block against transformers' SigLIP layer with the same weights; then an image request runs end to end on a
    checkpoint in the family's layout (decoder: a tiny Llama)."""
    import json
    import transformers as tf
    from safetensors.torch import save_file
    from vllm_distributed_amd.config import HFConfig
    from vllm_distributed_amd.models.deepseek_vl2 import MlpProjector, SiglipTimmTower, _TimmBlock
    torch.manual_seed(0)
    # block == transformers' SigLIP encoder layer
    sc = tf.SiglipVisionConfig(hidden_size=32, intermediate_size=64, num_attention_heads=2, num_hidden_layers=1,
                               image_size=28, patch_size=14)
    from transformers.models.siglip.modeling_siglip import SiglipEncoderLayer
    ref = SiglipEncoderLayer(sc).eval()
    blk = _TimmBlock(32, 2, 64).eval()
    sd = ref.state_dict()
    with torch.no_grad():
        blk.attn.qkv.weight.copy_(torch.cat([sd[f"self_attn.{x}_proj.weight"] for x in "qkv"]))
        blk.attn.qkv.bias.copy_(torch.cat([sd[f"self_attn.{x}_proj.bias"] for x in "qkv"]))
        blk.attn.proj.load_state_dict({"weight": sd["self_attn.out_proj.weight"], "bias": sd["self_attn.out_proj.bias"]})
        blk.norm1.load_state_dict({"weight": sd["layer_norm1.weight"], "bias": sd["layer_norm1.bias"]})
        blk.norm2.load_state_dict({"weight": sd["layer_norm2.weight"], "bias": sd["layer_norm2.bias"]})
        blk.mlp.fc1.load_state_dict({"weight": sd["mlp.fc1.weight"], "bias": sd["mlp.fc1.bias"]})
        blk.mlp.fc2.load_state_dict({"weight": sd["mlp.fc2.weight"], "bias": sd["mlp.fc2.bias"]})
        x = torch.randn(2, 4, 32)
        out = ref(x, None)
        torch.testing.assert_close(blk(x), out[0] if isinstance(out, tuple) else out, rtol=1e-4, atol=1e-5)
    # projector: neighbourhoods of a 3 x 3 grid (padded to 4 x 4), channel-major concatenation
    pc = HFConfig(dict(projector_type="downsample_mlp_gelu", input_dim=4, n_embed=8, depth=2, mlp_ratio=1,
                       downsample_ratio=2))
    proj = MlpProjector(pc).eval()
    f = torch.randn(1, 9, 4)
    grid = torch.zeros(4, 4, 4)
    grid[:3, :3] = f[0].view(3, 3, 4)
    cells = []
    for i in range(2):
        for j in range(2):
            nb = torch.stack([grid[2 * i, 2 * j], grid[2 * i, 2 * j + 1], grid[2 * i + 1, 2 * j], grid[2 * i + 1, 2 * j + 1]])
            cells.append(nb.t().reshape(-1))                                 # channel-major
    with torch.no_grad():
        torch.testing.assert_close(proj(f)[0], proj.layers(torch.stack(cells)))
    # end to end
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, head_dim=64,
                        tie_word_embeddings=False)
    hf = tf.LlamaForCausalLM(tc).float().eval()
    vc = dict(model_name="siglip", image_size=56, patch_size=14, width=32, layers=2, heads=2, mlp_ratio=2.0)
    pcd = dict(projector_type="downsample_mlp_gelu", input_dim=32, n_embed=128, depth=2, mlp_ratio=1,
               downsample_ratio=2)
    cfg = dict(architectures=["DeepseekVLV2ForCausalLM"], model_type="deepseek_vl_v2", vision_config=vc,
               projector_config=pcd, tile_tag="2D", global_view_pos="head", image_token_id=300,
               candidate_resolutions=[[56, 56], [56, 112], [112, 56], [112, 112]],
               language_config={**tc.to_dict(), "architectures": ["LlamaForCausalLM"]})
    path = os.path.join(str(tmp_path), "dsvl2")
    os.makedirs(path)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    tower, pr = SiglipTimmTower(HFConfig(vc)), MlpProjector(HFConfig(pcd))
    sd = {"language." + n: t for n, t in hf.state_dict().items()}
    sd.update({"vision." + n: torch.randn_like(t) * 0.2 for n, t in tower.state_dict().items()})
    sd.update({"projector." + n: torch.randn_like(t) * 0.2 for n, t in pr.state_dict().items()})
    sd["vision.attn_pool.latent"] = torch.zeros(1, 1, 32)                     # present in checkpoints, unused
    sd["image_newline"], sd["view_seperator"] = torch.randn(128), torch.randn(128)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    img = (np.random.default_rng(3).random((50, 110, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4)
    outs = llm.generate([{"prompt_token_ids": [5, 300, 9, 11], "multi_modal_data": {"image": img}},
                         {"prompt_token_ids": [5, 17, 9, 11]}],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    # best resolution for 50 x 110 is 56 x 112: two tiles across, one down; g = ceil(4 / 2) = 2
    assert list(outs[0].prompt_token_ids).count(300) == 2 * 3 + 1 + 2 * (2 * 2 + 1)
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    from vllm_distributed_amd.multimodal import process_multimodal
    _, inputs, _, _ = process_multimodal(llm.llm_engine.vllm_config, [5, 300, 9], {"image": img})
    px, crop = inputs[0]
    assert px.shape == (3, 3, 56, 56) and crop.tolist() == [2, 1]
    with torch.no_grad():
        got = model.encode_mm_items([_to(model, [px, crop])])[0]
        e = model.projector(model.vision(_to(model, px)))                    # [3, 4, 128]
        nl, sep = model.image_newline, model.view_seperator
        want = [e[0, 0], e[0, 1], nl, e[0, 2], e[0, 3], nl, sep,
                e[1, 0], e[1, 1], e[2, 0], e[2, 1], nl, e[1, 2], e[1, 3], e[2, 2], e[2, 3], nl]
        torch.testing.assert_close(got, torch.stack(want))
        ref = hf.generate(torch.tensor([[5, 17, 9, 11]]), max_new_tokens=6, do_sample=False, pad_token_id=0,
                          eos_token_id=None)[0, 4:].tolist()
    assert list(outs[1].outputs[0].token_ids) == ref and len(outs[0].outputs[0].token_ids) == 6
    llm.shutdown()


def test_kimi_vl_tower_and_engine(tmp_path):
    """Kimi-VL.  No transformers class exists (parity unpinned).  The 2-D rotary embedding is checked against complex
    multiplication with cis(column * f_i), cis(row * f_i) alternating, the patch merge against a loop, and an image of
    a size that needs padding runs end to end (decoder: a tiny Llama under ``language_model.``)."""
    import json
    import transformers as tf
    from safetensors.torch import save_file
    from vllm_distributed_amd.config import HFConfig
    from vllm_distributed_amd.models.kimi_vl import MoonViT, rope_2d_angles, rotate_pairs
    torch.manual_seed(0)
    h, w, D = 3, 5, 16
    x = torch.randn(h * w, 2, D)
    got = rotate_pairs(x, rope_2d_angles(h, w, D))
    f = 1.0 / 10000.0 ** (torch.arange(0, D, 4).float() / D)
    xc = torch.view_as_complex(x.view(h * w, 2, D // 2, 2).contiguous())
    want = torch.empty_like(xc)
    for r in range(h):
        for c in range(w):
            cis = torch.stack([torch.polar(torch.ones(D // 4), c * f), torch.polar(torch.ones(D // 4), r * f)], -1).reshape(-1)
            want[r * w + c] = xc[r * w + c] * cis
    torch.testing.assert_close(got, torch.view_as_real(want).flatten(-2), rtol=1e-5, atol=1e-5)
    vc = dict(hidden_size=32, num_attention_heads=2, intermediate_size=64, num_hidden_layers=2, patch_size=14,
              init_pos_emb_height=4, init_pos_emb_width=4, merge_kernel_size=[2, 2])
    tower = MoonViT(HFConfig(vc)).eval()
    with torch.no_grad():
        for p in tower.parameters():
            p.normal_(0, 0.2)
        patches = torch.randn(4 * 6 + 2 * 2, 3, 14, 14)
        outs = tower(patches, [(4, 6), (2, 2)])
        assert [tuple(o.shape) for o in outs] == [(6, 4, 32), (1, 4, 32)]
        alone = tower(patches[24:], [(2, 2)])[0]                             # images do not see each other
        torch.testing.assert_close(outs[1], alone, rtol=1e-4, atol=1e-5)
        # merged cell (1, 2) of the first image = patches (2, 4), (2, 5), (3, 4), (3, 5) of the 4 x 6 grid
        tower.merge = (1, 1)
        per_patch = tower(patches[:24], [(4, 6)])[0].view(4, 6, 32)
        tower.merge = (2, 2)
        torch.testing.assert_close(outs[0][1 * 3 + 2], torch.stack([per_patch[2, 4], per_patch[2, 5], per_patch[3, 4],
                                                                   per_patch[3, 5]]), rtol=1e-4, atol=1e-5)
    tc = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512, head_dim=64,
                        tie_word_embeddings=False)
    hf = tf.LlamaForCausalLM(tc).float().eval()
    cfg = dict(architectures=["KimiVLForConditionalGeneration"], model_type="kimi_vl", vision_config=vc,
               media_placeholder_token_id=300, text_config={**tc.to_dict(), "architectures": ["LlamaForCausalLM"]})
    path = os.path.join(str(tmp_path), "kimi")
    os.makedirs(path)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    sd = {"language_model." + n: t for n, t in hf.state_dict().items()}
    sd.update({"vision_tower." + n: t.detach().clone() for n, t in tower.state_dict().items()})
    sd.update({"multi_modal_projector.pre_norm.weight": torch.ones(32), "multi_modal_projector.pre_norm.bias": torch.zeros(32),
               "multi_modal_projector.linear_1.weight": torch.randn(128, 128) * 0.1,
               "multi_modal_projector.linear_1.bias": torch.zeros(128),
               "multi_modal_projector.linear_2.weight": torch.randn(128, 128) * 0.1,
               "multi_modal_projector.linear_2.bias": torch.zeros(128)})
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    img = (np.random.default_rng(3).random((50, 75, 3)) * 255).astype(np.uint8)       # 4 x 6 patches after padding
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4)
    outs = llm.generate([{"prompt_token_ids": [5, 300, 9, 11], "multi_modal_data": {"image": img}},
                         {"prompt_token_ids": [5, 17, 9, 11]}],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    assert list(outs[0].prompt_token_ids).count(300) == 2 * 3
    with torch.no_grad():
        ref = hf.generate(torch.tensor([[5, 17, 9, 11]]), max_new_tokens=6, do_sample=False, pad_token_id=0,
                          eos_token_id=None)[0, 4:].tolist()
    assert list(outs[1].outputs[0].token_ids) == ref and len(outs[0].outputs[0].token_ids) == 6
    llm.shutdown()


def test_minicpmv_slicing_and_engine(tmp_path):
    """MiniCPM-V 2.6.  No transformers class exists (parity unpinned).  The slice-grid rule on known cases, the
    resampler against its steps written out, the tower's bucketed position ids against transformers' Idefics2
    embedding, and an image large enough to be sliced end to end (decoder: Qwen2 from transformers under ``llm.``)."""
    import json
    import transformers as tf
    from safetensors.torch import save_file
    from vllm_distributed_amd.config import HFConfig
    from vllm_distributed_amd.models.minicpmv import MiniCPMVTower, best_resize, refine_size, slice_grid
    from vllm_distributed_amd.models.qwen_vl import QueryResampler, sincos_2d
    assert slice_grid((400, 300), 9, 448) is None
    assert slice_grid((1344, 896), 9, 448) == (3, 2)                         # 6 slices at a 3 : 2 aspect ratio
    assert slice_grid((4000, 500), 9, 448) == (8, 1)
    assert best_resize((1344, 896), 448, 14) == (546, 364)
    assert refine_size((1344, 896), (3, 2), 448, 14) == (3 * 448, 2 * 448)
    torch.manual_seed(0)
    # resampler: 5 queries without positions, keys with the table of their own 2 x 3 grid
    rs = QueryResampler(0, 128, 1, 32, post_projection=True, adaptive=True, num_queries=5).eval()
    with torch.no_grad():
        rs.query.normal_()
        rs.proj.normal_(0, 0.1)
        feats = torch.randn(6, 32)
        kv = rs.ln_kv(rs.kv_proj(feats))
        wq, wk, wv = rs.attn.in_proj_weight.chunk(3)
        bq, bk, bv = rs.attn.in_proj_bias.chunk(3)
        a = torch.softmax((rs.ln_q(rs.query) @ wq.t() + bq) @ ((kv + sincos_2d(128, 2, 3)) @ wk.t() + bk).t() / 128 ** 0.5, -1)
        want = rs.ln_post(rs.attn.out_proj(a @ (kv @ wv.t() + bv))) @ rs.proj
        torch.testing.assert_close(rs(feats[None], (2, 3))[0], want, rtol=1e-4, atol=1e-5)
    # tower position ids == transformers' Idefics2 embedding for a 2 x 3 slice on a 4 x 4 table
    vc = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, image_size=56,
              patch_size=14, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    tower = MiniCPMVTower(HFConfig(vc), drop_last=False).eval()
    from transformers.models.idefics2.modeling_idefics2 import Idefics2VisionEmbeddings
    emb = Idefics2VisionEmbeddings(tf.Idefics2VisionConfig(hidden_size=32, image_size=56, patch_size=14)).eval()
    with torch.no_grad():
        for p in tower.parameters():
            p.normal_(0, 0.2)
        emb.patch_embedding.load_state_dict(tower.embeddings.patch_embedding.state_dict())
        emb.position_embedding.load_state_dict(tower.embeddings.position_embedding.state_dict())
        px = torch.randn(3, 28, 42)
        ref = emb(px[None], torch.ones(1, 2, 3, dtype=torch.bool))[0]
        e = tower.embeddings
        mine = e.patch_embedding(px[None]).flatten(2).transpose(1, 2)[0] + e.position_embedding(tower._position_ids(2, 3, "cpu"))
        torch.testing.assert_close(mine, ref, rtol=1e-5, atol=1e-5)
    tc = tf.Qwen2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=1024,
                        tie_word_embeddings=False)
    hf = tf.Qwen2ForCausalLM(tc).float().eval()
    cfg = {**tc.to_dict(), "architectures": ["MiniCPMV"], "model_type": "minicpmv", "version": 2.6, "query_num": 5,
           "vision_config": vc, "drop_vision_last_layer": False, "image_size": 56, "patch_size": 14,
           "slice_config": {"max_slice_nums": 9, "scale_resolution": 56, "patch_size": 14}}
    path = os.path.join(str(tmp_path), "minicpmv")
    os.makedirs(path)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    sd = {"llm." + n: t for n, t in hf.state_dict().items()}
    sd.update({"vpm." + n: t.detach().clone() for n, t in tower.state_dict().items()})
    sd.update({"resampler." + n: t.detach().clone() for n, t in rs.state_dict().items()})
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    img = (np.random.default_rng(3).random((100, 150, 3)) * 255).astype(np.uint8)      # about 4.8 scale areas
    marks = {"im_start_id": 301, "im_end_id": 302, "slice_start_id": 303, "slice_end_id": 304, "unk_id": 300,
             "newline_id": 305}
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=4,
              mm_processor_kwargs=marks)
    outs = llm.generate([{"prompt_token_ids": [5, 301, 300, 302, 9, 11], "multi_modal_data": {"image": img}},
                         {"prompt_token_ids": [5, 17, 9, 11]}],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    ids = list(outs[0].prompt_token_ids)
    cols, rows = slice_grid((150, 100), 9, 56)
    assert (cols, rows) == (3, 2) and ids.count(300) == 5 * (1 + cols * rows)
    assert ids.count(303) == 6 and ids.count(305) == 1 and ids[1] == 301 and ids[7] == 302
    with torch.no_grad():
        ref = hf.generate(torch.tensor([[5, 17, 9, 11]]), max_new_tokens=6, do_sample=False, pad_token_id=0,
                          eos_token_id=None)[0, 4:].tolist()
    assert list(outs[1].outputs[0].token_ids) == ref and len(outs[0].outputs[0].token_ids) == 6
    llm.shutdown()


def test_minicpmo_audio_path_and_engine(tmp_path):
    """MiniCPM-o 2.6 (parity unpinned).  The audio path against transformers' Whisper encoder driven with the same
    chunk mask, then projector and pooling written out; a prompt with an image in front of an audio clip end to end."""
    import json
    import transformers as tf
    from safetensors.torch import save_file
    from vllm_distributed_amd.config import HFConfig
    from vllm_distributed_amd.models.minicpmv import MiniCPMVTower
    from vllm_distributed_amd.models.qwen_vl import QueryResampler
    torch.manual_seed(0)
    ac = tf.WhisperConfig(num_mel_bins=16, d_model=32, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=128,
                          max_source_positions=60, decoder_layers=1, decoder_attention_heads=2, decoder_ffn_dim=64,
                          vocab_size=100)
    from transformers.models.whisper.modeling_whisper import WhisperEncoder as HFEncoder
    enc = HFEncoder(ac).float().eval()
    vc = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, image_size=56,
              patch_size=14, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    tc = tf.Qwen2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=1024,
                        tie_word_embeddings=False)
    hf = tf.Qwen2ForCausalLM(tc).float().eval()
    cfg = {**tc.to_dict(), "architectures": ["MiniCPMO"], "model_type": "minicpmo", "version": 2.6, "query_num": 5,
           "vision_config": vc, "audio_config": ac.to_dict(), "drop_vision_last_layer": False, "image_size": 56,
           "patch_size": 14, "audio_pool_step": 2, "audio_chunk_length": 0.2,
           "slice_config": {"max_slice_nums": 9, "scale_resolution": 56, "patch_size": 14}}
    path = os.path.join(str(tmp_path), "minicpmo")
    os.makedirs(path)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    tower = MiniCPMVTower(HFConfig(vc), drop_last=False)
    rs = QueryResampler(0, 128, 1, 32, post_projection=True, adaptive=True, num_queries=5)
    sd = {"llm." + n: t for n, t in hf.state_dict().items()}
    sd.update({"vpm." + n: torch.randn_like(t) * 0.2 for n, t in tower.state_dict().items()})
    sd.update({"resampler." + n: torch.randn_like(t) * 0.2 for n, t in rs.state_dict().items()})
    sd.update({"apm." + n: t.detach().clone() for n, t in enc.state_dict().items()})
    proj = {"audio_projection_layer.linear1.weight": torch.randn(128, 32) * 0.2,
            "audio_projection_layer.linear1.bias": torch.randn(128) * 0.1,
            "audio_projection_layer.linear2.weight": torch.randn(128, 128) * 0.2,
            "audio_projection_layer.linear2.bias": torch.randn(128) * 0.1}
    sd.update(proj)
    sd["tts.some.weight"] = torch.zeros(2)                                    # the speech half is ignored
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    marks = {"im_start_id": 301, "im_end_id": 302, "slice_start_id": 303, "slice_end_id": 304, "unk_id": 300,
             "newline_id": 305, "audio_start_id": 306, "audio_end_id": 307}
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=4,
              mm_processor_kwargs=marks)
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    mel = torch.randn(16, 47, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        got = model.encode_audio(_to(model, mel)).cpu()
        # transformers' encoder wants max_source_positions * 2 mel frames: run its pieces by hand with the chunk mask
        x = torch.nn.functional.gelu(enc.conv1(mel[None]))
        x = torch.nn.functional.gelu(enc.conv2(x)).permute(0, 2, 1)
        S = x.shape[1]
        x = x + enc.embed_positions.weight[:S]
        idx = torch.arange(S)
        visible = idx[None, :] < ((idx // 10 + 1) * 10)[:, None]             # 0.2 s chunks = 10 frames
        m = torch.zeros(1, 1, S, S).masked_fill_(~visible, float("-inf"))
        for layer in enc.layers:
            out = layer(x, m)
            x = out[0] if isinstance(out, tuple) else out
        x = enc.layer_norm(x)
        y = torch.relu(x @ proj["audio_projection_layer.linear1.weight"].t() + proj["audio_projection_layer.linear1.bias"])
        y = y @ proj["audio_projection_layer.linear2.weight"].t() + proj["audio_projection_layer.linear2.bias"]
        want = torch.stack([(y[0, 2 * i] + y[0, 2 * i + 1]) / 2 for i in range((S - 2) // 2 + 1)])
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    img = (np.random.default_rng(3).random((40, 52, 3)) * 255).astype(np.uint8)
    out = llm.generate([{"prompt_token_ids": [5, 301, 300, 302, 9, 306, 300, 307, 11],
                         "multi_modal_data": {"image": img, "audio": {"audio_features": mel}}}],
                       SamplingParams(temperature=0, max_tokens=4, ignore_eos=True))[0]
    ids = list(out.prompt_token_ids)
    na = ((47 - 1) // 2 + 1 - 2) // 2 + 1
    assert ids == [5, 301] + [300] * 5 + [302, 9, 306] + [300] * na + [307, 11]
    assert len(out.outputs[0].token_ids) == 4
    llm.shutdown()


def test_ovis_soft_tokens_and_engine(tmp_path):
    """Ovis (parity unpinned).  The partition rule on known cases, the marker order, the 2 x 2 neighbourhood merge and
    the softmax-weighted lookup against loops, then an image that is partitioned end to end (decoder: Qwen2 under
    ``llm.``)."""
    import json
    import transformers as tf
    from safetensors.torch import save_file
    from vllm_distributed_amd.config import HFConfig
    from vllm_distributed_amd.models.ovis import VisualTokenizer, best_partition, marker_rows
    assert best_partition(448, 448, 448) == (1, 1)
    assert best_partition(896, 448, 448) == (1, 2)
    assert best_partition(448, 1344, 448) == (3, 1)
    V = 40
    assert marker_rows(1, (1, 1), V) == [39, 38, 35]
    assert marker_rows(5, (2, 2), V) == [39, 38, 37, 36, 37, 35]              # start, prefix, col, row, col, end
    torch.manual_seed(0)
    bc = dict(model_type="aimv2", hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
              image_size=56, patch_size=14, rms_norm_eps=1e-5, qkv_bias=False, use_bias=False, num_channels=3)
    vt = dict(backbone_config=bc, hidden_stride=2, vocab_size=V, tokenize_function="softmax", drop_cls_token=False)
    tokz = VisualTokenizer(HFConfig(vt)).eval()
    with torch.no_grad():
        for p in tokz.parameters():
            p.normal_(0, 0.3)
        px = torch.randn(2, 3, 56, 56)
        raw = tokz.backbone(px)                                              # [2, 16, 32], a 4 x 4 grid
        f = tokz.features(px)
        for i in range(2):
            for j in range(2):
                g = raw[0].view(4, 4, 32)
                want = torch.cat([g[2 * i, 2 * j], g[2 * i, 2 * j + 1], g[2 * i + 1, 2 * j], g[2 * i + 1, 2 * j + 1]])
                torch.testing.assert_close(f[0, i * 2 + j], want)
        probs = tokz(px)
        assert probs.shape == (2, 4, V) and torch.allclose(probs.sum(-1), torch.ones(2, 4)) and \
            float(probs[..., -5:].abs().max()) == 0.0
    tc = tf.Qwen2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=1024,
                        tie_word_embeddings=False)
    hf = tf.Qwen2ForCausalLM(tc).float().eval()
    cfg = dict(architectures=["Ovis"], model_type="ovis", visual_tokenizer_config=vt, image_pad_token_id=300,
               llm_config={**tc.to_dict(), "architectures": ["Qwen2ForCausalLM"]})
    path = os.path.join(str(tmp_path), "ovis")
    os.makedirs(path)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    sd = {"llm." + n: t for n, t in hf.state_dict().items()}
    sd.update({"visual_tokenizer." + n: t.detach().clone() for n, t in tokz.state_dict().items()})
    sd["visual_tokenizer.backbone.trunk.post_trunk_norm.weight"] = torch.ones(32)      # in checkpoints, unused
    sd["vte.weight"] = torch.randn(V, 128)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    img = (np.random.default_rng(3).random((56, 112, 3)) * 255).astype(np.uint8)        # partitioned 1 x 2
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4)
    outs = llm.generate([{"prompt_token_ids": [5, 300, 9, 11], "multi_modal_data": {"image": img}},
                         {"prompt_token_ids": [5, 17, 9, 11]}],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    assert best_partition(112, 56, 56) == (1, 2) and best_partition(115, 60, 56) == (2, 2)
    assert list(outs[0].prompt_token_ids).count(300) == 3 * 4 + 4             # overview + 2 crops, 4 markers
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    from vllm_distributed_amd.multimodal import process_multimodal
    _, inputs, _, _ = process_multimodal(llm.llm_engine.vllm_config, [5, 300, 9], {"image": img})
    with torch.no_grad():
        got = model.encode_mm_items([_to(model, inputs[0])])[0]
        words = model.visual_tokenizer(_to(model, inputs[0][0])) @ model.vte.weight
        w = model.vte.weight
        want = torch.cat([w[39:40], words[0], w[38:39], words[1], w[37:38], words[2], w[35:36]])
        torch.testing.assert_close(got, want)
        ref = hf.generate(torch.tensor([[5, 17, 9, 11]]), max_new_tokens=6, do_sample=False, pad_token_id=0,
                          eos_token_id=None)[0, 4:].tolist()
    assert list(outs[1].outputs[0].token_ids) == ref and len(outs[0].outputs[0].token_ids) == 6
    llm.shutdown()


def test_mllama_matches_hf(tmp_path):
    """Llama-3.2-Vision: tile tower with gated position embeddings and the "both padding" mask, intermediate taps,
    tanh-gated cross-attention layers fed from the per-request slot cache, a text-only request in the same batch.
    Image-first prompts (the family's chat format), where the original model and transformers agree on which rows see
    the image.  The tower taps the state ENTERING layer i as the original model does; transformers 5 collects the
    state LEAVING layer i, so the checkpoint config written for the engine carries transformers' indices + 1."""
    import json
    import transformers as tf
    torch.manual_seed(0)
    vc = dict(hidden_size=32, num_hidden_layers=4, num_global_layers=2, attention_heads=2, intermediate_size=64,
              vision_output_dim=32 * 3, image_size=28, patch_size=14, max_num_tiles=4, intermediate_layers_indices=[0, 2],
              supported_aspect_ratios=[[1, 1], [1, 2], [1, 3], [1, 4], [2, 1], [2, 2], [3, 1], [4, 1]])
    tc = dict(vocab_size=320, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1,
              intermediate_size=256, cross_attention_layers=[1, 3], max_position_embeddings=512, pad_token_id=0,
              bos_token_id=1, eos_token_id=2, rope_parameters={"rope_theta": 10000.0, "rope_type": "default"})
    cfg = tf.MllamaConfig(vision_config=vc, text_config=tc, image_token_index=320)
    hf = tf.MllamaForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)
            elif "gate" in n:
                p.fill_(0.7)
    hf = hf.float().eval()
    path = os.path.join(str(tmp_path), "mllama")
    hf.save_pretrained(path, safe_serialization=True)
    cj = json.load(open(os.path.join(path, "config.json")))
    cj["vision_config"]["intermediate_layers_indices"] = [1, 3]
    json.dump(cj, open(os.path.join(path, "config.json"), "w"))
    g = torch.Generator().manual_seed(5)
    px = torch.zeros(4, 3, 28, 28)
    px[:2] = torch.randn(2, 3, 28, 28, generator=g)                           # aspect ratio (1, 2): id 2, two tiles
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4,
              max_num_batched_tokens=64, block_size=16)
    prompts = [{"prompt_token_ids": [320, 1, 17, 45, 108],
                "multi_modal_data": {"image": {"pixel_values": px, "aspect_ratio_id": 2, "num_tiles": 2}}},
               {"prompt_token_ids": [1, 17, 45, 108, 9]}]
    outs = llm.generate(prompts, SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    llm.shutdown()
    with torch.no_grad():
        ids = torch.tensor([[320, 1, 17, 45, 108]])
        mask = torch.zeros(1, 5, 1, 4, dtype=torch.long)
        mask[:, :, 0, :2] = 1
        ref = hf.generate(input_ids=ids, pixel_values=px[None, None], aspect_ratio_ids=torch.tensor([[2]]),
                          aspect_ratio_mask=torch.tensor([[[1, 1, 0, 0]]]), cross_attention_mask=mask,
                          max_new_tokens=8, do_sample=False, eos_token_id=None, pad_token_id=0)[0, 5:].tolist()
        ref_text = hf.generate(input_ids=torch.tensor([[1, 17, 45, 108, 9]]), max_new_tokens=8, do_sample=False,
                               eos_token_id=None, pad_token_id=0)[0, 5:].tolist()
    assert list(outs[0].outputs[0].token_ids) == ref
    assert list(outs[1].outputs[0].token_ids) == ref_text


def _tiny_florence2(tmp_path):
    import transformers as tf
    torch.manual_seed(0)
    vc = dict(in_channels=3, depths=[1, 2], patch_size=[7, 3], patch_stride=[4, 2], patch_padding=[3, 1],
              patch_prenorm=[False, True], embed_dim=[32, 64], num_heads=[2, 4], num_groups=[2, 4], window_size=3,
              projection_dim=128, max_temporal_embeddings=10, max_position_embeddings=20, drop_path_rate=0.0)
    tc = dict(model_type="bart", vocab_size=200, d_model=128, encoder_layers=2, decoder_layers=2,
              encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=64, decoder_ffn_dim=64,
              max_position_embeddings=128, pad_token_id=1, bos_token_id=0, eos_token_id=2, decoder_start_token_id=2,
              forced_eos_token_id=None)
    cfg = tf.Florence2Config(vision_config=vc, text_config=tc, image_token_id=199)
    hf = tf.Florence2ForConditionalGeneration(cfg)
    with torch.no_grad():
        for p in hf.parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
    hf = hf.float().eval()
    path = os.path.join(str(tmp_path), "florence2")
    hf.save_pretrained(path, safe_serialization=True)
    return path, hf


def test_florence2_matches_hf(tmp_path):
    """Florence-2: DaViT tower (window attention with padding, channel-group attention, pre- and post-norm patch
    convolutions), position / frame embeddings and the two pooled sources, BART encoder over image vectors + prompt,
    decoder through the slot cache.  Greedy tokens against transformers, in its layout and in the original one."""
    import json
    from safetensors.torch import load_file, save_file
    path, hf = _tiny_florence2(tmp_path)
    g = torch.Generator().manual_seed(7)
    px = torch.randn(3, 40, 32, generator=g)                                  # 10 x 8 -> 5 x 4 positions (+ 1 pooled)
    text = [0, 17, 45, 108, 9, 2]
    with torch.no_grad():
        ref = hf.generate(input_ids=torch.tensor([[199] * 21 + text]), pixel_values=px[None], max_new_tokens=8,
                          do_sample=False, num_beams=1, eos_token_id=None, pad_token_id=1)[0, 1:].tolist()
        ref_text = hf.generate(input_ids=torch.tensor([text]), max_new_tokens=8, do_sample=False, num_beams=1,
                               eos_token_id=None, pad_token_id=1)[0, 1:].tolist()

    def run(model_path):
        llm = LLM(model=model_path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128,
                  max_num_seqs=4, block_size=16)
        outs = llm.generate([{"prompt_token_ids": text, "multi_modal_data": {"image": px}},
                             {"prompt_token_ids": text}], SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
        llm.shutdown()
        return [list(o.outputs[0].token_ids) for o in outs]
    assert run(path) == [ref, ref_text]

    # the original checkpoint layout (what the reference reads)
    sd = load_file(os.path.join(path, "model.safetensors"))
    swaps = ((".conv1.", ".conv1.fn.dw."), (".conv2.", ".conv2.fn.dw."), (".window_attn.", ".window_attn.fn."),
             (".channel_attn.", ".channel_attn.fn."), (".ffn.", ".ffn.fn.net."))
    old = {}
    for n, t in sd.items():
        n = n[len("model."):] if n.startswith("model.") else n
        if n.startswith("language_model."):
            n = "language_model.model." + n[len("language_model."):]
        elif n == "lm_head.weight":
            n = "language_model.lm_head.weight"
        elif n.startswith("vision_tower.blocks."):
            kind = "window_attn" if ".spatial_block." in n else "channel_attn"
            n = n.replace(".norm1.", f".{kind}.norm.").replace(".norm2.", ".ffn.norm.") if ".norm" in n else n
            if ".norm." not in n:
                for a, b in swaps:
                    n = n.replace(a, b)
        elif n.startswith("vision_tower.convs."):
            n = n.replace(".conv.", ".proj.")
        elif n == "multi_modal_projector.image_projection.weight":
            n, t = "image_projection", t.t().contiguous()
        elif n.startswith("multi_modal_projector.image_proj_norm."):
            n = n[len("multi_modal_projector."):]
        elif n.startswith("multi_modal_projector.image_position_embed."):
            n = "image_pos_embed." + n[len("multi_modal_projector.image_position_embed."):]
        elif n.endswith("pos_idx_to_embed"):
            n = "visual_temporal_embed.pos_idx_to_embed"
        old[n] = t.contiguous()
    path2 = os.path.join(str(tmp_path), "florence2_orig")
    os.makedirs(path2)
    save_file(old, os.path.join(path2, "model.safetensors"))
    cj = json.load(open(os.path.join(path, "config.json")))
    v = cj["vision_config"]
    v["dim_embed"] = v.pop("embed_dim")
    v["image_pos_embed"] = {"type": "learned_abs_2d", "max_pos_embeddings": v.pop("max_position_embeddings")}
    v["visual_temporal_embedding"] = {"type": "COSINE", "max_temporal_embeddings": v.pop("max_temporal_embeddings")}
    v["image_feature_source"] = ["spatial_avg_pool", "temporal_avg_pool"]
    cj["projection_dim"] = v["projection_dim"]
    json.dump(cj, open(os.path.join(path2, "config.json"), "w"))
    assert run(path2) == [ref, ref_text]


def test_keye_tower_projector_and_engine(tmp_path):
    """Keye-VL (parity unpinned): the NaViT SigLIP tower and the 2 x 2 projector against the arithmetic written out
    (separate q / k / v, complex rotations, per-head loops); then through the engine: a text-only request follows
    transformers' Qwen3, an image request equals the decoder run on the hand-assembled embeddings."""
    import json
    import math
    import torch.nn.functional as F
    import transformers as tf
    from safetensors.torch import save_file
    torch.manual_seed(0)
    tcfg = tf.Qwen3Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                          num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=512,
                          tie_word_embeddings=False,
                          rope_parameters={"rope_type": "default", "rope_theta": 10000.0, "mrope_section": [8, 12, 12]})
    lm = tf.Qwen3ForCausalLM(tcfg).float().eval()
    V, H, L, P, S = 32, 2, 2, 14, 4                                            # tower width, heads, layers, patch, table side
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g) * 0.3                          # noqa: E731
    sd = {"model." + k if not k.startswith("lm_head") else k: v for k, v in
          {k[len("model."):] if k.startswith("model.") else k: v for k, v in lm.state_dict().items()}.items()}
    vp = "visual.vision_model."
    sd.update({vp + "embeddings.patch_embedding.weight": r(V, 3, P, P), vp + "embeddings.patch_embedding.bias": r(V),
               vp + "embeddings.position_embedding.weight": r(S * S, V),
               vp + "embeddings.packing_position_embedding.weight": r(8, V),
               vp + "post_layernorm.weight": 1 + r(V) * 0.1, vp + "post_layernorm.bias": r(V) * 0.1,
               vp + "head.probe": r(1, 1, V),
               "mlp_AR.pre_norm.weight": 1 + r(V) * 0.1, "mlp_AR.pre_norm.bias": r(V) * 0.1,
               "mlp_AR.linear_1.weight": r(4 * V, 4 * V), "mlp_AR.linear_1.bias": r(4 * V),
               "mlp_AR.linear_2.weight": r(128, 4 * V), "mlp_AR.linear_2.bias": r(128)})
    for i in range(L):
        lp = f"{vp}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[lp + f"self_attn.{n}.weight"], sd[lp + f"self_attn.{n}.bias"] = r(V, V), r(V) * 0.1
        for n in ("layer_norm1", "layer_norm2"):
            sd[lp + n + ".weight"], sd[lp + n + ".bias"] = 1 + r(V) * 0.1, r(V) * 0.1
        sd[lp + "mlp.fc1.weight"], sd[lp + "mlp.fc1.bias"] = r(48, V), r(48) * 0.1
        sd[lp + "mlp.fc2.weight"], sd[lp + "mlp.fc2.bias"] = r(V, 48), r(V) * 0.1
    path = os.path.join(str(tmp_path), "keye")
    os.makedirs(path)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    cj = tcfg.to_dict()
    cj.update(architectures=["KeyeForConditionalGeneration"], model_type="Keye", image_token_id=300, video_token_id=301,
              rope_scaling={"type": "mrope", "mrope_section": [8, 12, 12]}, rope_theta=10000.0,
              vision_config=dict(hidden_size=V, num_attention_heads=H, num_hidden_layers=L, intermediate_size=48,
                                 patch_size=P, image_size=P * S, num_channels=3, hidden_act="gelu_pytorch_tanh",
                                 layer_norm_eps=1e-6))
    cj.pop("rope_parameters", None)
    json.dump(cj, open(os.path.join(path, "config.json"), "w"))

    def tower(px, t, h, w):                                                     # px [t * h * w, 3 * P * P] row-major
        x = px @ sd[vp + "embeddings.patch_embedding.weight"].reshape(V, -1).t() + sd[vp + "embeddings.patch_embedding.bias"]
        tab = sd[vp + "embeddings.position_embedding.weight"].view(1, S, S, V).permute(0, 3, 1, 2)
        pos = F.interpolate(tab, size=(h, w), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).reshape(h * w, V)
        x = x + pos.repeat(t, 1)
        hd = V // H
        freq = 1.0 / 10000.0 ** (torch.arange(hd // 4).float() * 2 / (hd // 2))
        N = t * h * w
        rot = torch.zeros(N, hd // 2, dtype=torch.complex64)
        for n in range(N):
            row, col = (n % (h * w)) // w, n % w
            rot[n] = torch.polar(torch.ones(hd // 2), torch.cat([row * freq, col * freq]))
        for i in range(L):
            lp = f"{vp}encoder.layers.{i}."
            hcur = F.layer_norm(x, (V,), sd[lp + "layer_norm1.weight"], sd[lp + "layer_norm1.bias"], 1e-6)
            heads = []
            for hh in range(H):
                sl = slice(hh * hd, (hh + 1) * hd)
                q, k, v = (hcur @ sd[lp + f"self_attn.{n}.weight"][sl].t() + sd[lp + f"self_attn.{n}.bias"][sl]
                           for n in ("q_proj", "k_proj", "v_proj"))
                def turn(a):                                                      # pairs (i, i + hd / 2)
                    c = torch.complex(a[:, : hd // 2], a[:, hd // 2:]) * rot
                    return torch.cat([c.real, c.imag], -1)
                att = torch.softmax(turn(q) @ turn(k).t() / math.sqrt(hd), -1)
                heads.append(att @ v)
            x = x + torch.cat(heads, -1) @ sd[lp + "self_attn.out_proj.weight"].t() + sd[lp + "self_attn.out_proj.bias"]
            hcur = F.layer_norm(x, (V,), sd[lp + "layer_norm2.weight"], sd[lp + "layer_norm2.bias"], 1e-6)
            hcur = F.gelu(hcur @ sd[lp + "mlp.fc1.weight"].t() + sd[lp + "mlp.fc1.bias"], approximate="tanh")
            x = x + hcur @ sd[lp + "mlp.fc2.weight"].t() + sd[lp + "mlp.fc2.bias"]
        x = F.layer_norm(x, (V,), sd[vp + "post_layernorm.weight"], sd[vp + "post_layernorm.bias"], 1e-6)
        x = F.layer_norm(x, (V,), sd["mlp_AR.pre_norm.weight"], sd["mlp_AR.pre_norm.bias"], 1e-5)
        out = []
        for tt in range(t):
            for a in range(h // 2):
                for b in range(w // 2):
                    idx = [tt * h * w + (2 * a + da) * w + 2 * b + db for da in range(2) for db in range(2)]
                    out.append(torch.cat([x[j] for j in idx]))
        y = torch.stack(out) @ sd["mlp_AR.linear_1.weight"].t() + sd["mlp_AR.linear_1.bias"]
        return F.gelu(y) @ sd["mlp_AR.linear_2.weight"].t() + sd["mlp_AR.linear_2.bias"]

    img = (torch.rand(56, 84, 3, generator=g) * 255).to(torch.uint8).numpy()    # already a multiple of 28: 4 x 6 patches
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4,
              block_size=16, mm_processor_kwargs={"min_pixels": 28 * 28})
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    proc = __import__("vllm_distributed_amd.multimodal", fromlist=["x"]).MULTIMODAL_REGISTRY.get_processor(
        llm.llm_engine.vllm_config)
    ids, inputs, _, ranges = proc([5, 300, 17, 45], {"image": img})
    assert ids == [5] + [300] * 6 + [17, 45] and (ranges[0].offset, ranges[0].length) == (1, 6)
    px, grid = inputs[0]
    assert grid.tolist() == [1, 4, 6] and px.shape == (24, 3 * P * P)
    t0 = torch.as_tensor(img).permute(2, 0, 1).float() / 255.0
    assert torch.allclose(px[7].view(3, P, P), ((t0 - 0.5) / 0.5)[:, P:2 * P, P:2 * P], atol=1e-6)   # row 1, column 1
    with torch.no_grad():
        got = model.encode_mm_items([[px, grid]])[0].cpu()
        want = tower(px, 1, 4, 6)
    torch.testing.assert_close(got, want, atol=2e-4, rtol=2e-4)
    vid = (torch.rand(2, 56, 56, 3, generator=g) * 255).to(torch.uint8).numpy()
    vids, vin, _, vr = proc([301, 9], {"video": vid})
    assert vin[0][1].tolist() == [2, 4, 4] and vr[0].length == 8
    with torch.no_grad():
        torch.testing.assert_close(model.encode_mm_items([vin[0]])[0].cpu(), tower(vin[0][0], 2, 4, 4), atol=2e-4,
                                   rtol=2e-4)

    # a 28 x 28 image is one token after the projector: its three position axes coincide, so transformers' Qwen3
    # (plain positions) serves as the decoder reference for the merged embeddings; multi-axis positions are the
    # inherited Qwen2-VL rule, tested with that family
    small = (torch.rand(28, 28, 3, generator=g) * 255).to(torch.uint8).numpy()
    sid, sin_, _, _ = proc([5, 300, 17, 45], {"image": small})
    assert sid == [5, 300, 17, 45] and sin_[0][1].tolist() == [1, 2, 2]
    outs = llm.generate([{"prompt_token_ids": [5, 300, 17, 45], "multi_modal_data": {"image": small}},
                         {"prompt_token_ids": [5, 9, 17, 45, 77]},
                         {"prompt_token_ids": [5, 300, 17, 45], "multi_modal_data": {"image": img}}],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    llm.shutdown()
    with torch.no_grad():
        ref_text = lm.generate(torch.tensor([[5, 9, 17, 45, 77]]), max_new_tokens=6, do_sample=False,
                               eos_token_id=None, pad_token_id=0)[0, 5:].tolist()
        emb = lm.model.embed_tokens(torch.tensor([sid]))
        emb[0, 1] = tower(sin_[0][0], 1, 2, 2)[0]
        ref_img = lm.generate(inputs_embeds=emb, max_new_tokens=6, do_sample=False, eos_token_id=None,
                              pad_token_id=0)[0].tolist()
    assert list(outs[1].outputs[0].token_ids) == ref_text
    assert list(outs[0].outputs[0].token_ids) == ref_img
    assert len(outs[2].outputs[0].token_ids) == 6


def test_molmo_backbone_layout_and_engine(tmp_path):
    """Molmo (parity unpinned): tower + 2 x 2 attention pooling + projector against the arithmetic written out; the
    crop / token layout against properties that follow from its definition (every pooled vector inside the used
    region is used exactly once, rows read across crops); the decoder (fused att_proj, [up | gate] ff_proj, 128
    extra embedding rows) against transformers' Qwen2 on a text-only prompt and on hand-assembled embeddings."""
    import json
    import math
    import torch.nn.functional as F
    import transformers as tf
    from safetensors.torch import save_file
    from vllm_distributed_amd.models.molmo import crops_and_layout, select_tiling
    torch.manual_seed(0)
    E, D, FF = 300, 128, 96
    tcfg = tf.Qwen2Config(vocab_size=E + 128, hidden_size=D, intermediate_size=FF, num_hidden_layers=2,
                          num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=2048,
                          tie_word_embeddings=False, rope_parameters={"rope_type": "default", "rope_theta": 10000.0})
    lm = tf.Qwen2ForCausalLM(tcfg).float().eval()
    hsd = lm.state_dict()
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g) * 0.2                          # noqa: E731
    V = dict(size=56, patch=14, dim=32, heads=2, layers=10, mlp=48, num_pos=17, eps=1e-5)    # 4 x 4 patches, 2 x 2 pooled
    sd = {"model.transformer.wte.embedding": hsd["model.embed_tokens.weight"][:E],
          "model.transformer.wte.new_embedding": hsd["model.embed_tokens.weight"][E:],
          "model.transformer.ln_f.weight": hsd["model.norm.weight"],
          "model.transformer.ff_out.weight": hsd["lm_head.weight"][:E]}
    for i in range(2):
        a, b = f"model.layers.{i}.", f"model.transformer.blocks.{i}."
        sd[b + "att_proj.weight"] = torch.cat([hsd[a + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        sd[b + "att_proj.bias"] = torch.cat([hsd[a + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        sd[b + "attn_out.weight"] = hsd[a + "self_attn.o_proj.weight"]
        sd[b + "ff_proj.weight"] = torch.cat([hsd[a + "mlp.up_proj.weight"], hsd[a + "mlp.gate_proj.weight"]], 0)
        sd[b + "ff_out.weight"] = hsd[a + "mlp.down_proj.weight"]
        sd[b + "attn_norm.weight"], sd[b + "ff_norm.weight"] = hsd[a + "input_layernorm.weight"], \
            hsd[a + "post_attention_layernorm.weight"]
    vb = "model.vision_backbone."
    d = V["dim"]
    sd.update({vb + "image_vit.class_embedding": r(d), vb + "image_vit.positional_embedding": r(17, d),
               vb + "image_vit.patch_embedding.weight": r(d, 588), vb + "image_vit.pre_ln.weight": 1 + r(d) * 0.1,
               vb + "image_vit.pre_ln.bias": r(d) * 0.1, vb + "pad_embed": r(2, 2 * d)})
    for i in range(V["layers"]):
        p = f"{vb}image_vit.transformer.resblocks.{i}."
        for n in ("wq", "wk", "wv", "wo"):
            sd[p + f"attention.{n}.weight"], sd[p + f"attention.{n}.bias"] = r(d, d), r(d) * 0.1
        sd[p + "feed_forward.w1.weight"], sd[p + "feed_forward.w1.bias"] = r(48, d), r(48) * 0.1
        sd[p + "feed_forward.w2.weight"], sd[p + "feed_forward.w2.bias"] = r(d, 48), r(d) * 0.1
        for n in ("attention_norm", "ffn_norm"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = 1 + r(d) * 0.1, r(d) * 0.1
    for n, shape in (("wq", (d, 2 * d)), ("wk", (d, 2 * d)), ("wv", (d, 2 * d)), ("wo", (d, d))):
        sd[vb + f"image_pooling_2d.{n}.weight"], sd[vb + f"image_pooling_2d.{n}.bias"] = r(*shape), r(d) * 0.1
    sd[vb + "image_projector.w1.weight"], sd[vb + "image_projector.w3.weight"] = r(FF, d), r(FF, d)
    sd[vb + "image_projector.w2.weight"] = r(D, FF)
    path = os.path.join(str(tmp_path), "molmo")
    os.makedirs(path)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    cj = dict(architectures=["MolmoForCausalLM"], model_type="molmo", vocab_size=E, embedding_size=E, hidden_size=D,
              intermediate_size=2 * FF, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
              max_position_embeddings=2048, rope_theta=10000.0, layer_norm_eps=1e-6, layer_norm_type="rms",
              qkv_bias=True, attention_layer_norm=False, norm_after=False, weight_tying=False, torch_dtype="float32",
              vision_backbone=V, image_start_token_id=E, image_end_token_id=E + 1, image_patch_token_id=E + 2,
              image_col_token_id=E + 3)
    json.dump(cj, open(os.path.join(path, "config.json"), "w"))

    # ---- layout -------------------------------------------------------------------------------------------------
    assert select_tiling(100, 100, 224, 12) == (1, 1) and select_tiling(300, 700, 224, 12) == (2, 4)
    assert select_tiling(5000, 5000, 224, 12) == (3, 4) or select_tiling(5000, 5000, 224, 12) == (3, 3)
    img = torch.rand(3, 500, 900, generator=g)
    patches, mask, order, rows, cols = crops_and_layout(img)                  # the released geometry: 336 / 14 / 4 + 4
    th, tw = select_tiling(500 - 112, 900 - 112, 224, 12)
    assert patches.shape == (1 + th * tw, 576, 588) and mask.shape == (1 + th * tw, 576)
    assert (rows, cols) == ((th * 16 + 8 + 1) // 2, (tw * 16 + 8 + 1) // 2) and order.numel() == 144 + rows * cols
    assert order[:144].tolist() == list(range(144)) and order.unique().numel() == order.numel()
    crop, posn = (order[144:] - 144) // 144, (order[144:] - 144) % 144
    grid_c = crop.view(rows, cols)
    assert (grid_c[0] // tw == 0).all() and (grid_c[-1] // tw == th - 1).all()       # top / bottom rows of crops
    assert (grid_c[:, 0] % tw == 0).all() and (grid_c[:, -1] % tw == tw - 1).all()
    # inside one crop the used block is contiguous; an inner crop starts 2 pooled positions in (its 4-patch margin)
    y, x = posn.view(rows, cols) // 12, posn.view(rows, cols) % 12
    assert int(x[0, 0]) == 0 and int(y[0, 0]) == 0
    if tw > 1:
        first_of_second = (grid_c[0] == 1).nonzero()[0, 0]
        assert int(x[0, first_of_second]) == 2 and int(first_of_second) == 10          # 4 + 16 patches -> 10 pooled
    assert mask[-1].eq(-1).all() and mask[:-1].min() >= 0 and mask[:-1].max() == 1

    # ---- tower, pooling, projector ----------------------------------------------------------------------------------
    def backbone(images, masks):
        T = images.shape[0]
        x = images @ sd[vb + "image_vit.patch_embedding.weight"].t()
        x = torch.cat([sd[vb + "image_vit.class_embedding"].expand(T, 1, d), x], 1) + sd[vb + "image_vit.positional_embedding"]
        x = F.layer_norm(x, (d,), sd[vb + "image_vit.pre_ln.weight"], sd[vb + "image_vit.pre_ln.bias"], 1e-5)

        def mha(pre, q_in, kv_in, heads=2):
            q, k, v = (t @ sd[pre + f"{n}.weight"].t() + sd[pre + f"{n}.bias"] for n, t in
                       (("wq", q_in), ("wk", kv_in), ("wv", kv_in)))
            outs = []
            for hh in range(heads):
                sl = slice(hh * d // heads, (hh + 1) * d // heads)
                att = torch.softmax(q[..., sl] @ k[..., sl].transpose(-1, -2) / math.sqrt(d // heads), -1)
                outs.append(att @ v[..., sl])
            return torch.cat(outs, -1) @ sd[pre + "wo.weight"].t() + sd[pre + "wo.bias"]
        states = []
        for i in range(V["layers"]):
            p = f"{vb}image_vit.transformer.resblocks.{i}."
            h = F.layer_norm(x, (d,), sd[p + "attention_norm.weight"], sd[p + "attention_norm.bias"], 1e-5)
            x = x + mha(p + "attention.", h, h)
            h = F.layer_norm(x, (d,), sd[p + "ffn_norm.weight"], sd[p + "ffn_norm.bias"], 1e-5)
            h = h @ sd[p + "feed_forward.w1.weight"].t() + sd[p + "feed_forward.w1.bias"]
            x = x + (h * torch.sigmoid(1.702 * h)) @ sd[p + "feed_forward.w2.weight"].t() + sd[p + "feed_forward.w2.bias"]
            states.append(x)
        f = torch.cat([states[-2], states[-9]], -1)[:, 1:]
        pe = sd[vb + "pad_embed"]
        f = f + pe[0] * (masks == 0).float()[..., None] + pe[1] * ((masks < 1) & (masks != 0)).float()[..., None]
        out = []
        for t in range(T):
            for a in range(2):
                for b in range(2):
                    blk = torch.stack([f[t, (2 * a + da) * 4 + 2 * b + db] for da in range(2) for db in range(2)])
                    out.append(mha(vb + "image_pooling_2d.", blk.mean(0, keepdim=True), blk)[0])
        o = torch.stack(out).view(T, 4, d)
        w1, w3, w2 = (sd[vb + f"image_projector.{n}.weight"] for n in ("w1", "w3", "w2"))
        return (F.silu(o @ w1.t()) * (o @ w3.t())) @ w2.t()

    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=4,
              block_size=16, mm_processor_kwargs={"max_crops": 4, "overlap_margins": [2, 0]})
    model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    small = torch.rand(3, 60, 100, generator=g)
    pt, mk, od, rr, cc = crops_and_layout(small, 4, 56, 14, (2, 0))
    with torch.no_grad():
        got = model.vision_backbone(pt, mk).cpu()
    torch.testing.assert_close(got, backbone(pt, mk), atol=2e-3, rtol=2e-3)

    # ---- engine -----------------------------------------------------------------------------------------------------
    text = [5, 17, 45, 108, 9]
    pre = {"images": pt, "image_masks": mk,
           "image_input_idx": torch.full((pt.shape[0], 4), -100, dtype=torch.long)}
    # the checkpoint's own processor output: vector k of crop c lands on prompt position image_input_idx[c, k]
    prompt = [5, E] + [E + 2, E + 2, E + 3] * 2 + [E + 1, 17, 45]
    patch_pos = [i for i, t in enumerate(prompt) if t == E + 2]
    pre["image_input_idx"][0] = torch.tensor(patch_pos)
    outs = llm.generate([{"prompt_token_ids": text},
                         {"prompt_token_ids": prompt, "multi_modal_data": {"image": pre}},
                         {"prompt_token_ids": [5, E + 2, 17, 45], "multi_modal_data": {"image": small}}],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    llm.shutdown()
    with torch.no_grad():
        def greedy(cur):                                                      # the head has only the first E rows
            toks = []
            for _ in range(6):
                toks.append(int(lm(inputs_embeds=cur).logits[0, -1, :E].argmax()))
                cur = torch.cat([cur, lm.model.embed_tokens(torch.tensor([[toks[-1]]]))], 1)
            return toks
        ref_text = greedy(lm.model.embed_tokens(torch.tensor([text])))
        emb = lm.model.embed_tokens(torch.tensor([prompt]))
        emb[0, patch_pos] = backbone(pt, mk)[0]
        ref_img = greedy(emb)
    assert list(outs[0].outputs[0].token_ids) == ref_text
    assert list(outs[1].outputs[0].token_ids) == ref_img
    # the raw image: brackets, rows closed by column tokens, vectors placed by ``order``
    full = [5, E] + ([E + 2] * 2 + [E + 3]) * 2 + [E + 1, E] + ([E + 2] * cc + [E + 3]) * rr + [E + 1, 17, 45]
    with torch.no_grad():
        emb = lm.model.embed_tokens(torch.tensor([full]))
        spots = [i for i, t in enumerate(full) if t == E + 2]
        emb[0, spots] = backbone(pt, mk).flatten(0, 1)[od]
        ref_raw = greedy(emb)
    assert list(outs[2].outputs[0].token_ids) == ref_raw


def test_molmo_norm_after_variant_follows_olmo2(tmp_path):
    """The OLMo-based Molmo decoder (norms behind attention and MLP, full-width q / k norms, no biases) is OLMo-2's
    block: a text-only prompt must follow transformers' Olmo2 with the weights renamed to Molmo's layout."""
    import json
    import transformers as tf
    from safetensors.torch import save_file
    torch.manual_seed(0)
    E, D, FF = 300, 128, 96
    cfg = tf.Olmo2Config(vocab_size=E + 128, hidden_size=D, intermediate_size=FF, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-6,
                         tie_word_embeddings=False, rope_parameters={"rope_type": "default", "rope_theta": 10000.0})
    lm = tf.Olmo2ForCausalLM(cfg).float().eval()
    with torch.no_grad():
        for n, p in lm.named_parameters():
            if "norm" in n:
                p.add_(torch.randn_like(p) * 0.1)
    h = lm.state_dict()
    sd = {"model.transformer.wte.embedding": h["model.embed_tokens.weight"][:E],
          "model.transformer.wte.new_embedding": h["model.embed_tokens.weight"][E:],
          "model.transformer.ln_f.weight": h["model.norm.weight"], "model.transformer.ff_out.weight": h["lm_head.weight"][:E]}
    for i in range(2):
        a, b = f"model.layers.{i}.", f"model.transformer.blocks.{i}."
        sd[b + "att_proj.weight"] = torch.cat([h[a + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        sd[b + "attn_out.weight"] = h[a + "self_attn.o_proj.weight"]
        sd[b + "q_norm.weight"], sd[b + "k_norm.weight"] = h[a + "self_attn.q_norm.weight"], h[a + "self_attn.k_norm.weight"]
        sd[b + "ff_proj.weight"] = torch.cat([h[a + "mlp.up_proj.weight"], h[a + "mlp.gate_proj.weight"]], 0)
        sd[b + "ff_out.weight"] = h[a + "mlp.down_proj.weight"]
        sd[b + "attn_norm.weight"] = h[a + "post_attention_layernorm.weight"]
        sd[b + "ff_norm.weight"] = h[a + "post_feedforward_layernorm.weight"]
    path = os.path.join(str(tmp_path), "molmo_o")
    os.makedirs(path)
    V = dict(size=56, patch=14, dim=32, heads=2, layers=10, mlp=48, num_pos=17, eps=1e-5)
    from vllm_distributed_amd.models.molmo import MolmoVisionBackbone
    sd.update({"model.vision_backbone." + k: v for k, v in MolmoVisionBackbone(D, FF, V).state_dict().items()})
    json.dump(dict(architectures=["MolmoForCausalLM"], model_type="molmo", vocab_size=E, embedding_size=E, hidden_size=D,
                   intermediate_size=2 * FF, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                   max_position_embeddings=512, rope_theta=10000.0, layer_norm_eps=1e-6, layer_norm_type="rms",
                   qkv_bias=False, attention_layer_norm=True, norm_after=True, weight_tying=False,
                   torch_dtype="float32", vision_backbone=V), open(os.path.join(path, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=2,
              block_size=16, load_format="auto")
    text = [5, 17, 45, 108, 9]
    out = llm.generate([{"prompt_token_ids": text}], SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))[0]
    llm.shutdown()
    with torch.no_grad():
        cur, ref = torch.tensor([text]), []
        for _ in range(6):
            ref.append(int(lm(cur).logits[0, -1, :E].argmax()))
            cur = torch.cat([cur, torch.tensor([[ref[-1]]])], 1)
    assert list(out.outputs[0].token_ids) == ref


def _tiny_phi4mm(tmp_path, name="phi4mm"):
    import transformers as tf
    torch.manual_seed(0)
    vc = dict(hidden_size=32, intermediate_size=48, num_hidden_layers=3, num_attention_heads=2, image_size=56,
              patch_size=14, crop_size=56, feature_layer=-2, image_token_id=300)
    ac = dict(hidden_size=32, intermediate_size=48, num_blocks=2, num_attention_heads=2, input_size=80,
              nemo_conv_channels=8, nemo_final_size=10, time_reduction=8, ext_pw_out_channel=32,
              depthwise_separable_out_channel=32, kernel_size=3, bias_max_distance=10, chunk_size=-1, left_chunk=18,
              downsample_rate=1, audio_token_id=301)
    cfg = tf.Phi4MultimodalConfig(vocab_size=320, hidden_size=128, intermediate_size=96, num_hidden_layers=2,
                                  num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=1024,
                                  original_max_position_embeddings=1024, vision_config=vc, audio_config=ac,
                                  pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False,
                                  rope_parameters={"rope_type": "default", "rope_theta": 10000.0,
                                                   "partial_rotary_factor": 0.75})
    hf = tf.Phi4MultimodalForCausalLM(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2 and "embed_tokens.weight" not in n:
                p.mul_(2.0)
            if "feature_extensor" in n or n.endswith((".b1", ".b2")):
                p.normal_(0, 0.5)
        enc = hf.model.embed_tokens_extend.audio_embed.encoder.encoder_embedding
        enc.global_mean.normal_(8, 1)
        enc.global_invstd.uniform_(0.2, 0.4)
    hf = hf.float().eval()
    path = os.path.join(str(tmp_path), name)
    hf.save_pretrained(path, safe_serialization=True)
    return path, hf


def test_phi4mm_matches_hf(tmp_path):
    """Phi-4-multimodal: masked NaViT SigLIP crops + HD arrangement, the conformer audio encoder (windows of 500 frames
    included), both projector pairs; greedy tokens against transformers for an image prompt with a partly padded
    canvas, a speech prompt, a prompt with both and a text prompt; the log-mel front end against transformers'
    feature extractor; the original checkpoint layout."""
    import json
    import transformers as tf
    from safetensors.torch import load_file, save_file
    from vllm_distributed_amd.models.phi4mm import hd_preprocess, image_token_count, log_mel_frames
    path, hf = _tiny_phi4mm(tmp_path)
    g = torch.Generator().manual_seed(3)
    img = (torch.rand(3, 80, 150, generator=g) * 255).round()                  # 2 x 3 crops, the bottom part padded
    px, size, masks = hd_preprocess(img, 56, 14, 36)
    assert size == (112, 168) and px.shape == (7, 3, 56, 56) and masks.shape == (7, 4, 4)
    # the image fills 89 of the canvas' 112 rows: one whole patch row of padding, masked in the bottom crops
    assert masks[0].all() and masks[1:4].all() and not masks[4:, 3, :].any() and masks[4:, :3, :].all()
    n_img = image_token_count(size, masks, 56)
    assert n_img == 4 * (6 + 1) + 1 + 2 * 3             # pooled rows 0 / 2 of the bottom crops count: 4 rows of 6; global 2 x 2
    wave = torch.randn(16000 * 3 // 2, generator=g) * 0.1
    feats = log_mel_frames(wave)
    fe = tf.Phi4MultimodalFeatureExtractor()
    want = fe(wave.numpy(), sampling_rate=16000, return_tensors="pt")
    torch.testing.assert_close(feats, want["audio_input_features"][0], atol=1e-4, rtol=1e-4)
    n_aud = int(want["audio_embed_sizes"][0])
    long_feats = torch.randn(4100, 80, generator=g) * 2 + 8                     # 513 frames after reduction: two windows

    text = [1, 17, 45, 108, 9]
    p_img, p_aud, p_both = [1, 300, 17, 45], [1, 301, 17, 45], [1, 300, 5, 301, 17]
    kw = dict(max_new_tokens=6, do_sample=False, eos_token_id=None, pad_token_id=0)
    ip = dict(image_pixel_values=px[None], image_sizes=torch.tensor([size]), image_attention_mask=masks[None].float())
    with torch.no_grad():
        r_text = hf.generate(torch.tensor([text]), **kw)[0, len(text):].tolist()
        ids = [1] + [300] * n_img + [17, 45]
        r_img = hf.generate(torch.tensor([ids]), **ip, **kw)[0, len(ids):].tolist()
        ids = [1] + [301] * n_aud + [17, 45]
        r_aud = hf.generate(torch.tensor([ids]), audio_input_features=feats[None], audio_embed_sizes=torch.tensor([n_aud]),
                            **kw)[0, len(ids):].tolist()
        ids = [1] + [301] * 513 + [17, 45]
        r_long = hf.generate(torch.tensor([ids]), audio_input_features=long_feats[None],
                             audio_embed_sizes=torch.tensor([513]), **kw)[0, len(ids):].tolist()
        ids = [1] + [300] * n_img + [5] + [301] * n_aud + [17]
        r_both = hf.generate(torch.tensor([ids]), **ip, audio_input_features=feats[None],
                             audio_embed_sizes=torch.tensor([n_aud]), **kw)[0, len(ids):].tolist()

    def run(model_path):
        llm = LLM(model=model_path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=1024,
                  max_num_seqs=4, block_size=16)
        outs = llm.generate([{"prompt_token_ids": text},
                             {"prompt_token_ids": p_img, "multi_modal_data": {"image": img}},
                             {"prompt_token_ids": p_aud, "multi_modal_data": {"audio": (wave.numpy(), 16000)}},
                             {"prompt_token_ids": p_aud, "multi_modal_data": {"audio": {"audio_input_features": long_feats}}},
                             {"prompt_token_ids": p_both, "multi_modal_data": {"image": img, "audio": (wave.numpy(), 16000)}}],
                            SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
        llm.shutdown()
        return [list(o.outputs[0].token_ids) for o in outs]
    assert run(path) == [r_text, r_img, r_aud, r_long, r_both]

    # the original layout: Sequential projections, glb_GN / sub_GN, base_layer in the decoder's linear names
    sd = load_file(os.path.join(path, "model.safetensors"))
    swaps = ((".img_projection_up.", ".img_projection.0."), (".img_projection_down.", ".img_projection.2."),
             (".up_proj_for_speech.", ".audio_projection.speech.0."), (".down_proj_for_speech.", ".audio_projection.speech.2."),
             (".up_proj_for_vision_speech.", ".audio_projection.vision.0."),
             (".down_proj_for_vision_speech.", ".audio_projection.vision.2."),
             (".global_img_feature_extensor", ".glb_GN"), (".sub_img_feature_extensor", ".sub_GN"))
    old = {}
    for n, t in sd.items():
        for a, b in swaps:
            n = n.replace(a, b)
        if n.startswith("model.layers.") and n.endswith("_proj.weight"):
            n = n.replace("_proj.weight", "_proj.base_layer.weight")
        old[n] = t.contiguous()
    path2 = os.path.join(str(tmp_path), "phi4mm_orig")
    os.makedirs(path2)
    save_file(old, os.path.join(path2, "model.safetensors"))
    cj = json.load(open(os.path.join(path, "config.json")))
    cj["architectures"] = ["Phi4MMForCausalLM"]
    json.dump(cj, open(os.path.join(path2, "config.json"), "w"))
    assert run(path2) == [r_text, r_img, r_aud, r_long, r_both]
