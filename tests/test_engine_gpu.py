"""End-to-end engine tests on the GPU: the HIP kernel path against Hugging Face fp32 logits,
hipGraph decode against eager decode, and sleep / wake-up on the address-stable allocator."""
import os

import pytest
import torch

from tests.test_e2e_cpu import _save_tiny

pytestmark = pytest.mark.gpu
PROMPTS = [[(i * 7 + j) % 300 + 1 for j in range(12 + 9 * i)] for i in range(6)]


def _llm(path, **kw):
    os.environ["VLLM_ENABLE_V1_MULTIPROCESSING"] = "0"
    from vllm_distributed_amd import LLM
    base = dict(skip_tokenizer_init=True, dtype="bfloat16", max_model_len=256, max_num_seqs=8,
                gpu_memory_utilization=0.05, seed=0)
    base.update(kw)
    return LLM(model=path, **base)


@pytest.mark.parametrize("kind", ["llama", "qwen2", "mixtral", "mamba", "mamba2", "deepseek_v2_mla", "bloom", "cohere",
                                  "gemma3", "olmo2", "falcon_new", "jamba", "bamba", "dbrx", "nemotron_h",
                                  "granitemoehybrid", "llama4", "minimax", "falcon_h1", "zamba2", "gemma3n"])
def test_gpu_logprobs_match_hf_and_graphs_match_eager(tmp_path, kind):
    import transformers as tf
    from vllm_distributed_amd import SamplingParams
    path, _ = _save_tiny(str(tmp_path), kind)
    sp = SamplingParams(temperature=0, max_tokens=12, ignore_eos=True, logprobs=0)
    # transformers' eager Mamba-2 normalises the gated output over the whole width, not per group
    extra = {"hf_overrides": {"gated_norm_groups": 1}} if kind in ("mamba2", "bamba", "granitemoehybrid") else {}
    a = _llm(path, **extra)                          # hipGraph decode
    outs_g = a.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
    if kind == "deepseek_v2_mla":      # the latent-cache kernel, not the expanded-head fallback, must be what ran
        mr = a.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
        assert all(getattr(m, "is_mla", False) for m in mr.attn_layers.values())
    a.shutdown()
    b = _llm(path, enforce_eager=True, **extra)
    outs_e = b.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
    b.shutdown()
    toks_g = [list(o.outputs[0].token_ids) for o in outs_g]
    toks_e = [list(o.outputs[0].token_ids) for o in outs_e]
    assert toks_g == toks_e, "graph replay and eager decode disagree"
    # teacher-forced check against HF in fp32: the log-probability the engine reports for each token
    # it generated must match HF's for the same context (bf16 tolerance)
    hf = tf.AutoModelForCausalLM.from_pretrained(path, torch_dtype=torch.float32).eval()
    worst = 0.0
    for p, o in zip(PROMPTS, outs_g):
        gen = list(o.outputs[0].token_ids)
        ids = torch.tensor([p + gen])
        with torch.no_grad():
            lp = torch.log_softmax(hf(ids).logits[0].float(), -1)
        for k, t in enumerate(gen):
            ref = float(lp[len(p) + k - 1, t])
            got = o.outputs[0].logprobs[k][t].logprob
            worst = max(worst, abs(ref - got))
    # Gemma-3n stacks four scaled streams, their mixing and a soft-capped head on every layer: transformers' own bf16
    # run of this tiny model is 0.125 away from its fp32 one
    assert worst < (0.2 if kind == "gemma3n" else 0.08), worst


def test_gpu_fp8_latent_cache_follows_hf(tmp_path):
    """DeepSeek-V2 with an fp8 latent cache: e4m3 rows in the MLA cache, log-probabilities of the generated tokens
    within fp8 tolerance of transformers' (teacher-forced)."""
    import transformers as tf
    from vllm_distributed_amd import SamplingParams
    path, _ = _save_tiny(str(tmp_path), "deepseek_v2_mla")
    llm = _llm(path, kv_cache_dtype="fp8")
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    assert all(getattr(m, "is_mla", False) and m.k_cache.dtype == torch.float8_e4m3fn for m in mr.attn_layers.values())
    outs = llm.generate(prompt_token_ids=PROMPTS,
                        sampling_params=SamplingParams(temperature=0, max_tokens=12, ignore_eos=True, logprobs=0))
    llm.shutdown()
    hf = tf.AutoModelForCausalLM.from_pretrained(path, torch_dtype=torch.float32).eval()
    worst = 0.0
    for p, o in zip(PROMPTS, outs):
        gen = list(o.outputs[0].token_ids)
        with torch.no_grad():
            lp = torch.log_softmax(hf(torch.tensor([p + gen])).logits[0].float(), -1)
        for k, t in enumerate(gen):
            worst = max(worst, abs(float(lp[len(p) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob))
    assert worst < 0.25, worst


@pytest.mark.parametrize("head_dim", [80, 96, 112, 192])
def test_gpu_padded_head_sizes_match_hf(tmp_path, head_dim):
    """Head sizes 96 and 192 run natively; 80 and 112 run zero-padded to 96 and 128.  Graphs on."""
    import transformers as tf
    from vllm_distributed_amd import SamplingParams
    torch.manual_seed(0)
    cfg = tf.LlamaConfig(vocab_size=320, hidden_size=2 * head_dim, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, head_dim=head_dim,
                         tie_word_embeddings=False)
    hf = tf.LlamaForCausalLM(cfg).float().eval()
    path = os.path.join(tmp_path, "m")
    hf.save_pretrained(path, safe_serialization=True)
    llm = _llm(path)
    outs = llm.generate(prompt_token_ids=PROMPTS,
                        sampling_params=SamplingParams(temperature=0, max_tokens=10, ignore_eos=True, logprobs=0))
    runner = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    rows = {m.k_cache.shape[-1] for m in runner.attn_layers.values()}
    llm.shutdown()
    assert rows == {{80: 96, 96: 96, 112: 128, 192: 192}[head_dim]}        # width of a cache row
    worst = 0.0
    for p, o in zip(PROMPTS, outs):
        gen = list(o.outputs[0].token_ids)
        with torch.no_grad():
            lp = torch.log_softmax(hf(torch.tensor([p + gen])).logits[0].float(), -1)
        for k, t in enumerate(gen):
            worst = max(worst, abs(float(lp[len(p) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob))
    assert worst < 0.08, worst


def test_sleep_allocator_keeps_addresses(tmp_path):
    from vllm_distributed_amd.device_allocator import SleepAllocator
    dev = torch.device("cuda:0")
    alloc = SleepAllocator.get()
    with alloc.use_memory_pool("weights"):
        w = torch.arange(64 << 20, device=dev, dtype=torch.int32)          # 256 MiB
    with alloc.use_memory_pool("kv_cache"):
        kv = torch.ones(128 << 20, device=dev, dtype=torch.uint8)
    ptr_w, ptr_kv = w.data_ptr(), kv.data_ptr()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    released = alloc.sleep(offload_tags=("weights",))
    free1 = torch.cuda.mem_get_info(dev)[0]
    assert released >= (256 << 20) + (128 << 20)
    assert free1 - free0 >= (300 << 20), (free0, free1)
    assert alloc.bytes("weights", 1) == 0 and alloc.bytes("weights", 2) >= (256 << 20)
    alloc.wake_up()
    assert w.data_ptr() == ptr_w and kv.data_ptr() == ptr_kv
    assert torch.equal(w[:1000].cpu(), torch.arange(1000, dtype=torch.int32))
    assert int(w[-1]) == (64 << 20) - 1
    kv.fill_(3)                                        # discarded pool: writable again
    assert int(kv.sum(dtype=torch.int64)) == 3 * (128 << 20)
    del w, kv


def test_engine_sleep_wake_with_graphs(tmp_path):
    """enable_sleep_mode: weights and KV cache live in the sleep allocator; after sleep(1) + wake_up
    the captured decode graphs still produce the same tokens."""
    from vllm_distributed_amd import SamplingParams
    path, _ = _save_tiny(str(tmp_path), "llama")
    sp = SamplingParams(temperature=0, max_tokens=10, ignore_eos=True)
    llm = _llm(path, enable_sleep_mode=True, enable_prefix_caching=False)
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    first = [list(o.outputs[0].token_ids) for o in llm.generate(prompt_token_ids=PROMPTS, sampling_params=sp)]
    free0 = torch.cuda.mem_get_info()[0]
    llm.sleep(1)
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 > free0, "sleep released nothing"
    llm.wake_up()
    again = [list(o.outputs[0].token_ids) for o in llm.generate(prompt_token_ids=PROMPTS, sampling_params=sp)]
    assert again == first
    assert mr.sleep_allocator is not None and mr.graphs, "expected the VMM path with captured graphs"
    llm.shutdown()


def test_sleep_forgets_cached_prefixes_and_clears_the_cache(tmp_path):
    """Sleeping through the engine core (the path the server's /sleep takes) with prefix caching on: the
    dropped KV cache must not serve prefix hits after wake-up, the re-backed cache memory must be clear
    (stale NaN patterns there would leak through zero-weighted slots), and the engine refuses to sleep
    while requests run."""
    from vllm_distributed_amd import SamplingParams
    path, _ = _save_tiny(str(tmp_path), "llama")
    sp = SamplingParams(temperature=0, max_tokens=10, ignore_eos=True)
    llm = _llm(path, enable_sleep_mode=True, enable_prefix_caching=True)
    core = llm.llm_engine.engine_core.engine_core
    mr = core.model_executor.worker.model_runner
    prompts = [p * 3 for p in PROMPTS]                   # several full blocks each
    first = [list(o.outputs[0].token_ids) for o in llm.generate(prompt_token_ids=prompts, sampling_params=sp)]
    # poison what the allocator will hand back: fill free device memory with NaN patterns, then release it
    core.sleep(1)
    junk = torch.full((256 << 20,), float("nan"), dtype=torch.float32, device="cuda")
    del junk
    torch.cuda.empty_cache()
    core.wake_up()
    assert all(float(k.float().abs().max()) == 0.0 and float(v.float().abs().max()) == 0.0
               for k, v in mr.kv_caches[:2]), "the KV cache came back with stale contents"
    again = [list(o.outputs[0].token_ids) for o in llm.generate(prompt_token_ids=prompts, sampling_params=sp)]
    assert again == first
    # a running request blocks sleep
    llm.llm_engine.add_request("busy", {"prompt_token_ids": prompts[0]}, sp)
    llm.llm_engine.step()
    with pytest.raises(RuntimeError, match="while requests are running"):
        core.sleep(1)
    llm.llm_engine.abort_request(["busy"])
    llm.shutdown()


def test_lora_fused_kernels_match_grouped_gemms(tmp_path, monkeypatch):
    """Multi-LoRA on the GPU: the fused shrink / expand kernels (decode-sized batches) and the grouped
    library-GEMM path must generate the same tokens, adapters must change the output, and a batch may mix
    two adapters with base-model requests."""
    from tests.test_lora_cpu import PROMPTS as LP, _make_adapter
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.request import LoRARequest
    path, hf = _save_tiny(str(tmp_path), "llama")
    a1, _ = _make_adapter(str(tmp_path), "ad1", hf, 1)
    a2, _ = _make_adapter(str(tmp_path), "ad2", hf, 2, targets=("k_proj", "o_proj", "up_proj"))
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
    reqs = [LoRARequest("ad1", 1, a1), None, LoRARequest("ad2", 2, a2)]
    runs = {}
    for name, limit in (("fused", "1024"), ("grouped", "0")):
        monkeypatch.setenv("VDA_LORA_KERNEL_MAX_TOKENS", limit)
        llm = _llm(path, enable_lora=True, max_loras=2, max_cpu_loras=2, max_lora_rank=8)
        runs[name] = [list(o.outputs[0].token_ids)
                      for o in llm.generate(prompt_token_ids=LP, sampling_params=sp, lora_request=reqs)]
        if name == "fused":
            runs["base"] = [list(o.outputs[0].token_ids) for o in llm.generate(prompt_token_ids=LP, sampling_params=sp)]
        llm.shutdown()
    assert runs["fused"][1] == runs["base"][1]                       # the request without an adapter is untouched
    assert runs["fused"][0] != runs["base"][0] and runs["fused"][2] != runs["base"][2]
    same = sum(a == b for x, y in zip(runs["fused"], runs["grouped"]) for a, b in zip(x, y))
    assert same >= 0.8 * sum(len(x) for x in runs["fused"]), (runs["fused"], runs["grouped"])


def test_cascade_attention_on_gpu(tmp_path, monkeypatch):
    """Shared cached prefix: the cascade path (bidirectional prefix pass + shifted suffix pass + merge
    kernels) against the ordinary kernels, first-token log-probabilities within bf16 tolerance."""
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.layers.attention import Attention
    monkeypatch.setenv("VDA_CASCADE_MIN_PREFIX", "32")
    monkeypatch.setenv("VDA_CASCADE_MIN_SEQS", "2")
    path, _ = _save_tiny(str(tmp_path), "llama")
    prefix = [(i * 13) % 300 + 1 for i in range(150)]
    prompts = [prefix + [5 + i, 9, 11 + 2 * i] * (i + 1) for i in range(6)]
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, logprobs=0)
    calls = []
    orig = Attention._cascade
    monkeypatch.setattr(Attention, "_cascade", lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1])
    res = {}
    for name, off in (("cascade", False), ("plain", True)):
        llm = _llm(path, enforce_eager=True, enable_prefix_caching=True, disable_cascade_attn=off)
        llm.generate(prompt_token_ids=[prefix + [1, 2]], sampling_params=SamplingParams(temperature=0, max_tokens=1))
        n0 = len(calls)
        res[name] = llm.generate(prompt_token_ids=prompts, sampling_params=sp)
        res[name + "_calls"] = len(calls) - n0
        llm.shutdown()
    assert res["cascade_calls"] > 0 and res["plain_calls"] == 0
    same = 0
    for a, b in zip(res["cascade"], res["plain"]):
        ta, tb = list(a.outputs[0].token_ids), list(b.outputs[0].token_ids)
        same += ta == tb
        if ta[0] == tb[0]:
            assert abs(a.outputs[0].logprobs[0][ta[0]].logprob - b.outputs[0].logprobs[0][tb[0]].logprob) < 0.05
    assert same >= 4


def test_qwen2_vl_on_gpu(tmp_path):
    """Image prompt on the GPU: vision tower, embedding merge, multimodal rotary rows through the fused
    RoPE + cache-write kernel in prefill, offset positions in graph-captured decode.  Teacher-forced
    log-probabilities against transformers in fp32."""
    from tests.test_multimodal_cpu import _tiny_qwen2_vl
    from vllm_distributed_amd import SamplingParams
    path, hf = _tiny_qwen2_vl(str(tmp_path))
    g = torch.Generator().manual_seed(3)
    img = {"pixel_values": torch.randn(4 * 6, 1176, generator=g), "image_grid_thw": torch.tensor([1, 4, 6])}
    p = [5, 17, 302, 300, 303, 9, 11, 250]
    ids = [5, 17, 302] + [300] * 6 + [303, 9, 11, 250]
    sp = SamplingParams(temperature=0, max_tokens=10, ignore_eos=True, logprobs=0)
    llm = _llm(path, enable_prefix_caching=False)
    outs = llm.generate([{"prompt_token_ids": p, "multi_modal_data": {"image": [img]}},
                         {"prompt_token_ids": [7, 8, 9, 10] * 4}], sp)
    llm.shutdown()
    for o, full, kw in ((outs[0], ids, dict(pixel_values=img["pixel_values"], image_grid_thw=img["image_grid_thw"][None],
                                            mm_token_type_ids=None)),
                        (outs[1], [7, 8, 9, 10] * 4, {})):
        gen = list(o.outputs[0].token_ids)
        seq = full + gen
        if kw:
            kw["mm_token_type_ids"] = torch.tensor([[1 if t == 300 else 0 for t in seq]], dtype=torch.int32)
        with torch.no_grad():
            lp = torch.log_softmax(hf(input_ids=torch.tensor([seq]), **kw).logits[0].float(), -1)
        worst = max(abs(float(lp[len(full) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob)
                    for k, t in enumerate(gen))
        assert worst < 0.08, worst


def test_encoder_only_embeddings_on_gpu(tmp_path):
    """BERT on the GPU: bidirectional mode of the paged prefill kernel, [CLS] embeddings against transformers."""
    from tests.test_pooling_cpu import _tiny_encoder
    path, hf = _tiny_encoder(str(tmp_path), "bert")
    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 30, [2, 3]]
    llm = _llm(path, max_model_len=128)
    outs = llm.embed(None, prompt_token_ids=prompts)
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = torch.nn.functional.normalize(hf(torch.tensor([p])).last_hidden_state[0, 0], dim=-1)
        got = torch.as_tensor(o.outputs.embedding).float()
        assert float(torch.dot(got, ref)) > 0.995, float(torch.dot(got, ref))


def test_whisper_and_qwen2_audio_on_gpu(tmp_path):
    """Audio models on the GPU: Whisper's cross-attention runs through the paged MFMA kernels over the
    per-slot cross cache (hipGraph decode included); log-probabilities against transformers in fp32."""
    import numpy as np
    import transformers as tf
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.multimodal import AudioTokenProcessor
    torch.manual_seed(0)
    cfg = tf.WhisperConfig(vocab_size=320, num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2,
                           decoder_layers=2, decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256,
                           max_source_positions=1500, max_target_positions=64, pad_token_id=0, bos_token_id=1,
                           eos_token_id=2, decoder_start_token_id=3, suppress_tokens=None, begin_suppress_tokens=None)
    hf = tf.WhisperForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)
    hf = hf.float().eval()
    path = os.path.join(tmp_path, "whisper")
    hf.save_pretrained(path, safe_serialization=True)
    rng = np.random.default_rng(5)
    clips = [rng.standard_normal(n).astype(np.float32) * 0.2 for n in (48000, 20000, 5000, 30000)]
    dec_prompts = [[3, 10, 11, 12], [3, 10], [3, 20, 21, 22, 23, 24], [3]]
    llm = _llm(path, max_model_len=64)
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    reqs = [{"prompt_token_ids": p, "multi_modal_data": {"audio": c}} for p, c in zip(dec_prompts, clips)]
    outs = llm.generate(reqs, SamplingParams(temperature=0, max_tokens=10, ignore_eos=True, logprobs=0))
    assert mr.graphs, "decode graphs expected"
    llm.shutdown()
    proc = AudioTokenProcessor(-1, 80)
    worst = 0.0
    for p, c, o in zip(dec_prompts, clips, outs):
        gen = list(o.outputs[0].token_ids)
        feats, _ = proc.features(c)
        with torch.no_grad():
            lg = hf(input_features=feats.unsqueeze(0), decoder_input_ids=torch.tensor([p + gen])).logits[0]
        lp = torch.log_softmax(lg.float(), -1)
        for k, t in enumerate(gen):
            worst = max(worst, abs(float(lp[len(p) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob))
    assert worst < 0.1, worst


@pytest.mark.parametrize("orig_max,head_dim", [(0, 128), (40, 64)])
def test_gpu_dual_chunk_attention_matches_dense_definition(tmp_path, orig_max, head_dim):
    """Dual-chunk attention through the paged prefill kernels (three passes per step, merged by log-sum-exp;
    virtual sequences without older chunks carry zero keys), graphs against eager, and the reported
    log-probabilities against the dense fp32 form of the definition."""
    from vllm_distributed_amd import SamplingParams
    from tests.test_dual_chunk_cpu import dense_dual_chunk_logits, save_tiny_dual_chunk
    path, hf = save_tiny_dual_chunk(str(tmp_path), orig_max, head_dim)
    g = torch.Generator().manual_seed(1)
    prompts = [torch.randint(3, 300, (n,), generator=g).tolist() for n in (150, 37, 101, 46)]
    sp = SamplingParams(temperature=0, max_tokens=12, ignore_eos=True, logprobs=0)
    a = _llm(path, max_model_len=512, enable_prefix_caching=False)
    mr = a.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    assert mr.dual_chunk and mr.graphs
    outs_g = a.generate(prompt_token_ids=prompts, sampling_params=sp)
    a.shutdown()
    b = _llm(path, max_model_len=512, enable_prefix_caching=False, enforce_eager=True)
    outs_e = b.generate(prompt_token_ids=prompts, sampling_params=sp)
    b.shutdown()
    assert [list(o.outputs[0].token_ids) for o in outs_g] == [list(o.outputs[0].token_ids) for o in outs_e]
    worst = 0.0
    for p, o in zip(prompts, outs_g):
        ids = list(p)
        with torch.no_grad():
            for k, t in enumerate(o.outputs[0].token_ids):
                lp = torch.log_softmax(dense_dual_chunk_logits(hf, ids, 64, 16, orig_max, len(p)).float(), -1)
                worst = max(worst, abs(float(lp[t]) - o.outputs[0].logprobs[k][t].logprob))
                ids.append(t)
    assert worst < 0.08, worst


def test_gpu_eagle3_speculative_tokens_equal_plain_greedy(tmp_path):
    """EAGLE-3 on the GPU with decode graphs on: the target writes its three-depth residual stream into the
    persistent buffer inside the captured graphs, the draft runs through the paged kernels, and greedy output
    equals the run without speculation (fp32-exact argmax is not required of bf16: compare with the same engine
    configuration minus the draft)."""
    import json
    from safetensors.torch import save_file
    from vllm_distributed_amd import SamplingParams
    path, _ = _save_tiny(str(tmp_path), "llama")
    cfg = json.load(open(os.path.join(path, "config.json")))
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    hd, nh, nkv = cfg["head_dim"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g) * 0.05      # noqa: E731
    w = {"fc.weight": r(H, 3 * H), "midlayer.self_attn.q_proj.weight": r(nh * hd, 2 * H),
         "midlayer.self_attn.k_proj.weight": r(nkv * hd, 2 * H), "midlayer.self_attn.v_proj.weight": r(nkv * hd, 2 * H),
         "midlayer.self_attn.o_proj.weight": r(H, nh * hd), "midlayer.mlp.gate_proj.weight": r(I, H),
         "midlayer.mlp.up_proj.weight": r(I, H), "midlayer.mlp.down_proj.weight": r(H, I),
         "midlayer.input_layernorm.weight": torch.ones(H), "midlayer.hidden_norm.weight": torch.ones(H),
         "midlayer.post_attention_layernorm.weight": torch.ones(H), "norm.weight": torch.ones(H),
         "lm_head.weight": r(V, H)}
    d = os.path.join(str(tmp_path), "eagle3")
    os.makedirs(d)
    cfg.update(architectures=["LlamaForCausalLMEagle3"], num_hidden_layers=1)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in w.items()}, os.path.join(d, "model.safetensors"))
    sp = SamplingParams(temperature=0, max_tokens=16, ignore_eos=True)
    a = _llm(path)
    ref = [list(o.outputs[0].token_ids) for o in a.generate(prompt_token_ids=PROMPTS, sampling_params=sp)]
    a.shutdown()
    b = _llm(path, speculative_config={"method": "eagle3", "model": d, "num_speculative_tokens": 3})
    mr = b.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    assert mr.model.model.aux_hidden is not None
    got = [list(o.outputs[0].token_ids) for o in b.generate(prompt_token_ids=PROMPTS, sampling_params=sp)]
    b.shutdown()
    same = sum(x == y for x, y in zip(got, ref))
    # verification runs the target on several tokens per step (other kernels than one-token decode): allow one
    # sequence to part ways at a bf16 near-tie
    assert same >= len(ref) - 1, (got, ref)


def test_mllama_cross_attention_on_gpu(tmp_path):
    """Llama-3.2-Vision on the GPU: the cross-attention layers read the per-request slot cache through the paged
    MFMA kernels (decode graphs included), an image request and a text-only request side by side; log-probabilities
    against transformers in fp32."""
    import json
    import transformers as tf
    from vllm_distributed_amd import SamplingParams
    torch.manual_seed(0)
    vc = dict(hidden_size=32, num_hidden_layers=4, num_global_layers=2, attention_heads=2, intermediate_size=64,
              vision_output_dim=32 * 3, image_size=28, patch_size=14, max_num_tiles=4, intermediate_layers_indices=[0, 2],
              supported_aspect_ratios=[[1, 1], [1, 2], [1, 3], [1, 4], [2, 1], [2, 2], [3, 1], [4, 1]])
    tc = dict(vocab_size=320, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1,
              intermediate_size=256, cross_attention_layers=[1, 3], max_position_embeddings=512, pad_token_id=0,
              bos_token_id=1, eos_token_id=2, rope_parameters={"rope_theta": 10000.0, "rope_type": "default"})
    hf = tf.MllamaForConditionalGeneration(tf.MllamaConfig(vision_config=vc, text_config=tc, image_token_index=320))
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if p.dim() >= 2:
                p.mul_(2.0)
            elif "gate" in n:
                p.fill_(0.7)
    hf = hf.float().eval()
    path = os.path.join(tmp_path, "mllama")
    hf.save_pretrained(path, safe_serialization=True)
    cj = json.load(open(os.path.join(path, "config.json")))
    cj["vision_config"]["intermediate_layers_indices"] = [1, 3]        # state entering layer i (see the CPU test)
    json.dump(cj, open(os.path.join(path, "config.json"), "w"))
    px = torch.zeros(4, 3, 28, 28)
    px[:2] = torch.randn(2, 3, 28, 28, generator=torch.Generator().manual_seed(5))
    llm = _llm(path, max_model_len=128)
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    prompts = [{"prompt_token_ids": [320, 1, 17, 45, 108],
                "multi_modal_data": {"image": {"pixel_values": px, "aspect_ratio_id": 2, "num_tiles": 2}}},
               {"prompt_token_ids": [1, 17, 45, 108, 9]}]
    outs = llm.generate(prompts, SamplingParams(temperature=0, max_tokens=10, ignore_eos=True, logprobs=0))
    assert mr.graphs, "decode graphs expected"
    llm.shutdown()
    mask = torch.zeros(1, 15, 1, 4, dtype=torch.long)
    mask[:, :, 0, :2] = 1
    worst = 0.0
    for k, o in enumerate(outs):
        p, gen = prompts[k]["prompt_token_ids"], list(o.outputs[0].token_ids)
        kw = dict(pixel_values=px[None, None], aspect_ratio_ids=torch.tensor([[2]]),
                  aspect_ratio_mask=torch.tensor([[[1, 1, 0, 0]]]), cross_attention_mask=mask) if k == 0 else {}
        with torch.no_grad():
            lp = torch.log_softmax(hf(input_ids=torch.tensor([p + gen]), **kw).logits[0].float(), -1)
        for j, t in enumerate(gen):
            worst = max(worst, abs(float(lp[len(p) + j - 1, t]) - o.outputs[0].logprobs[j][t].logprob))
    assert worst < 0.1, worst


def test_phi3_small_blocksparse_on_gpu(tmp_path):
    """Phi-3-small on the GPU: prompts inside the local span take the MFMA kernels, a longer one the masked PyTorch
    form over the paged cache; log-probabilities against the dense fp32 definition of the CPU test."""
    import json
    from safetensors.torch import save_file
    from tests.test_phi3_small_cpu import CFG, _dense_forward, _weights
    from vllm_distributed_amd import SamplingParams
    path = os.path.join(tmp_path, "phi3small")
    os.makedirs(path)
    json.dump(CFG, open(os.path.join(path, "config.json"), "w"))
    sd = _weights()
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    prompts = [[5, 17, 99, 3, 250], [(7 * i + 3) % 300 for i in range(37)]]
    llm = _llm(path, max_model_len=128, max_num_batched_tokens=16, enable_prefix_caching=False)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=6,
                                                                                   ignore_eos=True, logprobs=0))
    llm.shutdown()
    worst = 0.0
    for p, o in zip(prompts, outs):
        ids = list(p)
        for k, t in enumerate(o.outputs[0].token_ids):
            lp = torch.log_softmax(_dense_forward(sd, ids).float(), -1)
            worst = max(worst, abs(float(lp[t]) - o.outputs[0].logprobs[k][t].logprob))
            ids.append(t)
    assert worst < 0.1, worst


def test_plamo2_hybrid_on_gpu(tmp_path):
    """PLaMo 2 on the GPU: the HIP convolution and scan kernels with per-head time steps computed from the convolved
    signal, chunked prefill, hipGraph decode; log-probabilities against the fp32 recurrence of the CPU test."""
    import json
    from safetensors.torch import save_file
    from tests.test_plamo2_cpu import CFG, _dense_forward, _weights
    from vllm_distributed_amd import SamplingParams
    path = os.path.join(tmp_path, "plamo2")
    os.makedirs(path)
    json.dump(CFG, open(os.path.join(path, "config.json"), "w"))
    sd = _weights()
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    prompts = [[5, 17, 99, 3, 250], [(7 * i + 3) % 300 for i in range(37)]]
    llm = _llm(path, max_model_len=128, max_num_batched_tokens=16)
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=SamplingParams(temperature=0, max_tokens=6,
                                                                                   ignore_eos=True, logprobs=0))
    llm.shutdown()
    worst = 0.0
    for p, o in zip(prompts, outs):
        ids = list(p)
        for k, t in enumerate(o.outputs[0].token_ids):
            lp = torch.log_softmax(_dense_forward(sd, ids).float(), -1)
            worst = max(worst, abs(float(lp[t]) - o.outputs[0].logprobs[k][t].logprob))
            ids.append(t)
    assert worst < 0.1, worst


def test_gpu_sparse_dual_chunk_prefill(tmp_path):
    """Vertical-slash sparse prefill of dual-chunk attention on the GPU: with a budget that covers every key the
    tokens follow the dense definition; with a small one they follow the selection written out row by row."""
    from tests.test_dual_chunk_cpu import _with_sparse, dense_dual_chunk_logits, save_tiny_dual_chunk
    from vllm_distributed_amd import SamplingParams
    g = torch.Generator().manual_seed(3)
    p = torch.randint(3, 300, (500,), generator=g).tolist()
    for k, (budget, oracle) in enumerate((((4096, 4096), None), ((34, 104), (34, 104, 8)))):
        path, hf = save_tiny_dual_chunk(os.path.join(tmp_path, str(k)), 0)
        _with_sparse(path, *budget, False)
        llm = _llm(path, max_model_len=512, max_num_seqs=1, enable_prefix_caching=False, max_num_batched_tokens=512)
        out = llm.generate([{"prompt_token_ids": p}],
                           SamplingParams(temperature=0, max_tokens=3, ignore_eos=True, logprobs=0))[0]
        llm.shutdown()
        ids, worst = list(p), 0.0
        with torch.no_grad():
            for j, tok in enumerate(out.outputs[0].token_ids):
                lp = torch.log_softmax(dense_dual_chunk_logits(hf, ids, 64, 16, 0, len(p), sparse_budget=oracle), -1)
                worst = max(worst, abs(float(lp[tok]) - out.outputs[0].logprobs[j][tok].logprob))
                ids.append(tok)
        assert worst < 0.1, (budget, worst)


def test_gpu_florence2_and_phi4mm_follow_hf(tmp_path):
    """The towers written in plain PyTorch (DaViT, masked SigLIP crops, conformer) on the GPU in bf16, their vectors
    through the HIP decoder path: teacher-forced log-probabilities against transformers in fp32."""
    from tests.test_multimodal_cpu import _tiny_florence2, _tiny_phi4mm
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.models.phi4mm import hd_preprocess, image_token_count, log_mel_frames
    sp = SamplingParams(temperature=0, max_tokens=6, ignore_eos=True, logprobs=0)
    g = torch.Generator().manual_seed(7)

    # Florence-2 (encoder-decoder: the decoder starts from its start token alone, so the request's first step has the
    # shape of a decode step; it must not be served by a captured graph, which cannot hold the cross-attention write)
    path, hf = _tiny_florence2(os.path.join(tmp_path, "f"))
    px, text = torch.randn(3, 40, 32, generator=g), [0, 17, 45, 108, 9, 2]
    llm = _llm(path, max_model_len=128, max_num_seqs=2)
    assert llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.graphs
    out = llm.generate([{"prompt_token_ids": text, "multi_modal_data": {"image": px}}], sp)[0].outputs[0]
    llm.shutdown()
    gen = list(out.token_ids)
    with torch.no_grad():
        lg = hf(input_ids=torch.tensor([[199] * 21 + text]), pixel_values=px[None],
                decoder_input_ids=torch.tensor([[2] + gen[:-1]])).logits[0]
    lp = torch.log_softmax(lg.float(), -1)
    worst = max(abs(float(lp[k, t]) - out.logprobs[k][t].logprob) for k, t in enumerate(gen))
    assert worst < 0.1, ("florence2", worst)

    # Phi-4-multimodal: image + speech in one prompt
    path, hf = _tiny_phi4mm(os.path.join(tmp_path, "p"))
    img = (torch.rand(3, 80, 150, generator=g) * 255).round()
    wave = torch.randn(16000, generator=g) * 0.1
    pxs, size, masks = hd_preprocess(img, 56, 14, 36)
    feats = log_mel_frames(wave)
    n_img, n_aud = image_token_count(size, masks, 56), -(-feats.shape[0] // 8)
    llm = _llm(path, max_model_len=512, max_num_seqs=2)
    out = llm.generate([{"prompt_token_ids": [1, 300, 5, 301, 17],
                         "multi_modal_data": {"image": img, "audio": (wave.numpy(), 16000)}}], sp)[0].outputs[0]
    llm.shutdown()
    gen = list(out.token_ids)
    ids = [1] + [300] * n_img + [5] + [301] * n_aud + [17]
    with torch.no_grad():
        lg = hf(input_ids=torch.tensor([ids + gen[:-1]]), image_pixel_values=pxs[None], image_sizes=torch.tensor([size]),
                image_attention_mask=masks[None].float(), audio_input_features=feats[None],
                audio_embed_sizes=torch.tensor([n_aud])).logits[0, len(ids) - 1:]
    lp = torch.log_softmax(lg.float(), -1)
    worst = max(abs(float(lp[k, t]) - out.logprobs[k][t].logprob) for k, t in enumerate(gen))
    assert worst < 0.1, ("phi4mm", worst)


def test_gpu_in_flight_nf4_follows_hf_with_rounded_weights(tmp_path):
    """quantization="bitsandbytes" on the GPU: the projections live as NF4 codes + fp32 block scales and run the
    code-book kernels (skinny for decode, expansion for the prompt); log-probabilities follow transformers holding
    the NF4-rounded weights, graph replay equals eager decode."""
    import transformers as tf
    from vllm_distributed_amd import SamplingParams, ops
    from vllm_distributed_amd.layers.quant import nf4_dequantize, nf4_quantize
    path, _ = _save_tiny(str(tmp_path), "llama")
    sp = SamplingParams(temperature=0, max_tokens=12, ignore_eos=True, logprobs=0)
    calls = []
    real = ops.cb4_linear
    ops.cb4_linear = lambda x, *a, **k: (calls.append(x.shape[0]), real(x, *a, **k))[1]
    try:
        a = _llm(path, quantization="bitsandbytes")
        model = a.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
        qkv = model.model.layers[0].self_attn.qkv_proj
        assert qkv.qweight.dtype == torch.uint8 and qkv.qweight.is_cuda and getattr(qkv, "weight", None) is None
        outs_g = a.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
        a.shutdown()
        b = _llm(path, quantization="bitsandbytes", enforce_eager=True)
        outs_e = b.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
        b.shutdown()
    finally:
        ops.cb4_linear = real
    assert any(m <= 64 for m in calls) and any(m > 64 for m in calls)
    assert [list(o.outputs[0].token_ids) for o in outs_g] == [list(o.outputs[0].token_ids) for o in outs_e]
    hf = tf.AutoModelForCausalLM.from_pretrained(path, torch_dtype=torch.float32).eval()
    with torch.no_grad():
        for name, mod in hf.named_modules():
            if isinstance(mod, torch.nn.Linear) and name != "lm_head":
                mod.weight.copy_(nf4_dequantize(*nf4_quantize(mod.weight.data), torch.float32))
    worst = 0.0
    for p, o in zip(PROMPTS, outs_g):
        gen = list(o.outputs[0].token_ids)
        with torch.no_grad():
            lp = torch.log_softmax(hf(torch.tensor([p + gen])).logits[0].float(), -1)
        for k, t in enumerate(gen):
            worst = max(worst, abs(float(lp[len(p) + k - 1, t]) - o.outputs[0].logprobs[k][t].logprob))
    assert worst < 0.08, worst
