"""Architecture name -> model class (reference: vllm/model_executor/models/registry.py).
Classes are imported lazily; plugins may add entries with ``ModelRegistry.register_model``."""
from __future__ import annotations

import importlib
from typing import Type, Union

_MODELS: dict[str, tuple[str, str]] = {
    "LlamaForCausalLM": ("llama", "LlamaForCausalLM"),
    "LLaMAForCausalLM": ("llama", "LlamaForCausalLM"),
    "MistralForCausalLM": ("llama", "MistralForCausalLM"),
    "Qwen2ForCausalLM": ("llama", "Qwen2ForCausalLM"),
    "Qwen3ForCausalLM": ("llama", "Qwen3ForCausalLM"),
    "MixtralForCausalLM": ("mixtral", "MixtralForCausalLM"),
    "DeciLMForCausalLM": ("llama_variants", "DeciLMForCausalLM"),
    "Grok1ModelForCausalLM": ("mixtral", "Grok1ModelForCausalLM"),
    "Grok1ForCausalLM": ("mixtral", "Grok1ModelForCausalLM"),
    "Llama4ForCausalLM": ("llama4", "Llama4ForCausalLM"),
    "Llama4ForConditionalGeneration": ("llama4", "Llama4ForConditionalGeneration"),
    "QuantMixtralForCausalLM": ("mixtral", "MixtralForCausalLM"),
    "InternLMForCausalLM": ("llama", "LlamaForCausalLM"),
    "Ernie4_5ForCausalLM": ("more_decoders", "Ernie4_5ForCausalLM"),
    "Ernie4_5_ForCausalLM": ("more_decoders", "Ernie4_5ForCausalLM"),
    "Ernie4_5_MoeForCausalLM": ("more_decoders", "Ernie4_5_MoeForCausalLM"),
    "Dots1ForCausalLM": ("more_decoders", "Dots1ForCausalLM"),
    "HunYuanDenseV1ForCausalLM": ("more_decoders", "HunYuanDenseV1ForCausalLM"),
    "HunYuanMoEV1ForCausalLM": ("more_decoders", "HunYuanMoEV1ForCausalLM"),
    "MiMoForCausalLM": ("more_decoders", "MiMoForCausalLM"),
    "MiniCPMForCausalLM": ("more_decoders", "MiniCPMForCausalLM"),
    "PersimmonForCausalLM": ("more_decoders", "PersimmonForCausalLM"),
    "MiniCPM3ForCausalLM": ("more_decoders", "MiniCPM3ForCausalLM"),
    "ExaoneForCausalLM": ("renamed_decoders", "ExaoneForCausalLM"),
    "TeleChat2ForCausalLM": ("renamed_decoders", "TeleChat2ForCausalLM"),
    "TeleFLMForCausalLM": ("renamed_decoders", "TeleFLMForCausalLM"),
    "OrionForCausalLM": ("renamed_decoders", "OrionForCausalLM"),
    "QWenLMHeadModel": ("renamed_decoders", "QWenLMHeadModel"),
    "ChatGLMModel": ("renamed_decoders", "ChatGLMModel"),
    "ChatGLMForConditionalGeneration": ("renamed_decoders", "ChatGLMForConditionalGeneration"),
    "GritLM": ("renamed_decoders", "GritLM"),
    "Qwen2MoeForCausalLM": ("mixtral", "Qwen2MoeForCausalLM"),
    "Qwen3MoeForCausalLM": ("mixtral", "Qwen3MoeForCausalLM"),
    "PhiMoEForCausalLM": ("mixtral", "PhiMoEForCausalLM"),
    "PhimoeForCausalLM": ("mixtral", "PhiMoEForCausalLM"),
    "DbrxForCausalLM": ("mixtral", "DbrxForCausalLM"),
    "OlmoeForCausalLM": ("mixtral", "OlmoeForCausalLM"),
    "GraniteMoeForCausalLM": ("mixtral", "GraniteMoeForCausalLM"),
    "GemmaForCausalLM": ("llama_family", "GemmaForCausalLM"),
    "Gemma2ForCausalLM": ("llama_family", "Gemma2ForCausalLM"),
    "Phi3ForCausalLM": ("llama_family", "Phi3ForCausalLM"),
    "GraniteForCausalLM": ("llama_family", "GraniteForCausalLM"),
    "StableLmForCausalLM": ("llama_variants", "StableLmForCausalLM"),
    "StableLMEpochForCausalLM": ("llama_variants", "StableLmForCausalLM"),
    "OlmoForCausalLM": ("llama_variants", "OlmoForCausalLM"),
    "Olmo2ForCausalLM": ("llama_variants", "Olmo2ForCausalLM"),
    "CohereForCausalLM": ("llama_variants", "CohereForCausalLM"),
    "Cohere2ForCausalLM": ("llama_variants", "Cohere2ForCausalLM"),
    "GraniteMoeSharedForCausalLM": ("mixtral", "GraniteMoeSharedForCausalLM"),
    "NemotronForCausalLM": ("llama_variants", "NemotronForCausalLM"),
    "GlmForCausalLM": ("llama_variants", "GlmForCausalLM"),
    "Glm4ForCausalLM": ("llama_variants", "Glm4ForCausalLM"),
    "Gemma3ForCausalLM": ("llama_variants", "Gemma3ForCausalLM"),
    "Gemma3ForConditionalGeneration": ("llama_variants", "Gemma3TextOnlyFromMultimodal"),
    "InternLM2ForCausalLM": ("llama_variants", "InternLM2ForCausalLM"),
    "InternLM2VEForCausalLM": ("llama_variants", "InternLM2VEForCausalLM"),
    "InternLM3ForCausalLM": ("llama", "LlamaForCausalLM"),
    "BaichuanForCausalLM": ("llama_variants", "BaichuanForCausalLM"),
    "BaiChuanForCausalLM": ("llama_variants", "BaichuanForCausalLM"),
    "AquilaForCausalLM": ("llama", "LlamaForCausalLM"),
    "AquilaModel": ("llama", "LlamaForCausalLM"),
    "XverseForCausalLM": ("llama", "LlamaForCausalLM"),
    "SolarForCausalLM": ("llama", "LlamaForCausalLM"),
    "FalconForCausalLM": ("classic_decoders", "FalconForCausalLM"),
    "RWForCausalLM": ("classic_decoders", "FalconForCausalLM"),
    "BloomForCausalLM": ("classic_decoders", "BloomForCausalLM"),
    "MptForCausalLM": ("classic_decoders", "MptForCausalLM"),
    "MPTForCausalLM": ("classic_decoders", "MptForCausalLM"),
    "PhiForCausalLM": ("classic_decoders", "PhiForCausalLM"),
    "GPTNeoXForCausalLM": ("gpt_family", "GPTNeoXForCausalLM"),
    "GPTJForCausalLM": ("gpt_family", "GPTJForCausalLM"),
    "Starcoder2ForCausalLM": ("gpt_family", "Starcoder2ForCausalLM"),
    "GPTBigCodeForCausalLM": ("gpt_family", "GPTBigCodeForCausalLM"),
    "DeepseekV2ForCausalLM": ("deepseek", "DeepseekV2ForCausalLM"),
    "DeepseekV3ForCausalLM": ("deepseek", "DeepseekV3ForCausalLM"),
    "DeepseekForCausalLM": ("deepseek", "DeepseekV2ForCausalLM"),
    "Qwen2VLForConditionalGeneration": ("qwen2_vl", "Qwen2VLForConditionalGeneration"),
    "Qwen2_5_VLForConditionalGeneration": ("qwen2_vl", "Qwen2_5_VLForConditionalGeneration"),
    "LlavaForConditionalGeneration": ("llava", "LlavaForConditionalGeneration"),
    "MantisForConditionalGeneration": ("llava", "LlavaForConditionalGeneration"),
    "Qwen2AudioForConditionalGeneration": ("qwen2_audio", "Qwen2AudioForConditionalGeneration"),
    "WhisperForConditionalGeneration": ("whisper", "WhisperForConditionalGeneration"),
    "JambaForCausalLM": ("jamba", "JambaForCausalLM"),
    "BambaForCausalLM": ("jamba", "BambaForCausalLM"),
    "Zamba2ForCausalLM": ("zamba2", "Zamba2ForCausalLM"),
    "Gemma3nForConditionalGeneration": ("gemma3n", "Gemma3nForConditionalGeneration"),
    "Gemma3nForCausalLM": ("gemma3n", "Gemma3nForConditionalGeneration"),
    "NemotronHForCausalLM": ("jamba", "NemotronHForCausalLM"),
    "GraniteMoeHybridForCausalLM": ("jamba", "GraniteMoeHybridForCausalLM"),
    "MambaForCausalLM": ("mamba", "MambaForCausalLM"),
    "FalconMambaForCausalLM": ("mamba", "FalconMambaForCausalLM"),
    "Mamba2ForCausalLM": ("mamba", "Mamba2ForCausalLM"),
    "OPTForCausalLM": ("opt", "OPTForCausalLM"),
    "GPT2LMHeadModel": ("opt", "GPT2LMHeadModel"),
    "JAISLMHeadModel": ("opt", "JAISLMHeadModel"),
    "BertModel": ("bert", "BertEmbeddingModel"),
    "BertForMaskedLM": ("bert", "BertEmbeddingModel"),
    "BertForSequenceClassification": ("bert", "BertForSequenceClassification"),
    "RobertaModel": ("bert", "RobertaEmbeddingModel"),
    "RobertaForMaskedLM": ("bert", "RobertaEmbeddingModel"),
    "XLMRobertaModel": ("bert", "RobertaEmbeddingModel"),
    "RobertaForSequenceClassification": ("bert", "RobertaForSequenceClassification"),
    "XLMRobertaForSequenceClassification": ("bert", "RobertaForSequenceClassification"),
    "LlamaModel": ("pooling", "LlamaEmbeddingModel"),
    "Qwen2ForRewardModel": ("pooling", "Qwen2ForRewardModel"),
    "Qwen2ForProcessRewardModel": ("pooling", "Qwen2ForProcessRewardModel"),
    "MistralModel": ("pooling", "LlamaEmbeddingModel"),
    "Qwen2Model": ("pooling", "LlamaEmbeddingModel"),
    "Qwen3Model": ("pooling", "LlamaEmbeddingModel"),
    "MistralForSequenceClassification": ("pooling", "LlamaForSequenceClassification"),
    "Qwen3ForSequenceClassification": ("pooling", "LlamaForSequenceClassification"),
    "LlamaForSequenceClassification": ("pooling", "LlamaForSequenceClassification"),
    "Qwen2ForSequenceClassification": ("pooling", "LlamaForSequenceClassification"),
    "GPT2ForSequenceClassification": ("pooling", "GPT2ForSequenceClassification"),
    "GemmaForSequenceClassification": ("pooling", "GemmaForSequenceClassification"),
    "JambaForSequenceClassification": ("pooling", "JambaForSequenceClassification"),
    "Gemma2Model": ("pooling", "Gemma2EmbeddingModel"),
    "InternLM2ForRewardModel": ("pooling", "InternLM2ForRewardModel"),
    "EagleLlamaForCausalLM": ("eagle", "EagleLlamaForCausalLM"),
    "EagleMiniCPMForCausalLM": ("eagle", "EagleMiniCPMForCausalLM"),
    "LlamaForCausalLMEagle": ("eagle", "EagleLlamaForCausalLM"),
    "LlavaNextForConditionalGeneration": ("llava_next", "LlavaNextForConditionalGeneration"),
    "LlavaOnevisionForConditionalGeneration": ("llava_next", "LlavaOnevisionForConditionalGeneration"),
    "FuyuForCausalLM": ("fuyu", "FuyuForCausalLM"),
    "MedusaModel": ("eagle", "MedusaModel"),
    "LlavaNextVideoForConditionalGeneration": ("llava_next", "LlavaNextVideoForConditionalGeneration"),
    "NomicBertModel": ("bert", "NomicBertModel"),
    "GteNewModel": ("bert", "GteNewModel"),
    "GteModel": ("bert", "SnowflakeGteNewModel"),
    "ModernBertModel": ("bert", "ModernBertModel"),
    "ModernBertForSequenceClassification": ("bert", "ModernBertForSequenceClassification"),
    "TarsierForConditionalGeneration": ("llava", "TarsierForConditionalGeneration"),
    "PaliGemmaForConditionalGeneration": ("paligemma", "PaliGemmaForConditionalGeneration"),
    "ChameleonForConditionalGeneration": ("chameleon", "ChameleonForConditionalGeneration"),
    "Blip2ForConditionalGeneration": ("blip2", "Blip2ForConditionalGeneration"),
    "AriaForConditionalGeneration": ("aria", "AriaForConditionalGeneration"),
    "GraniteSpeechForConditionalGeneration": ("granite_speech", "GraniteSpeechForConditionalGeneration"),
    "Glm4vForConditionalGeneration": ("glm4v", "Glm4vForConditionalGeneration"),
    "QwenVLForConditionalGeneration": ("qwen_vl", "QwenVLForConditionalGeneration"),
    "GLM4VForCausalLM": ("glm4v_chat", "GLM4VForCausalLM"),
    "Qwen2_5OmniForConditionalGeneration": ("qwen2_5_omni", "Qwen2_5OmniThinkerForConditionalGeneration"),
    "Qwen2_5OmniModel": ("qwen2_5_omni", "Qwen2_5OmniThinkerForConditionalGeneration"),
    "Qwen2_5OmniThinkerForConditionalGeneration": ("qwen2_5_omni", "Qwen2_5OmniThinkerForConditionalGeneration"),
    "Tarsier2ForConditionalGeneration": ("qwen2_vl", "Qwen2VLForConditionalGeneration"),
    "BartForConditionalGeneration": ("bart", "BartForConditionalGeneration"),
    "BartModel": ("bart", "BartModel"),
    "Fairseq2LlamaForCausalLM": ("llama_variants", "Fairseq2LlamaForCausalLM"),
    "InternVLChatModel": ("internvl", "InternVLChatModel"),
    "NVLM_D": ("internvl", "NVLM_D_Model"),
    "ArcticForCausalLM": ("mixtral", "ArcticForCausalLM"),
    "MiniMaxVL01ForConditionalGeneration": ("llava_next", "MiniMaxVL01ForConditionalGeneration"),
    "Eagle3LlamaForCausalLM": ("eagle", "Eagle3LlamaForCausalLM"),
    "LlamaForCausalLMEagle3": ("eagle", "Eagle3LlamaForCausalLM"),
    "UltravoxModel": ("ultravox", "UltravoxModel"),
    "FalconH1ForCausalLM": ("jamba", "FalconH1ForCausalLM"),
    "MiniMaxForCausalLM": ("minimax", "MiniMaxForCausalLM"),
    "MiniMaxText01ForCausalLM": ("minimax", "MiniMaxText01ForCausalLM"),
    "MiniMaxM1ForCausalLM": ("minimax", "MiniMaxM1ForCausalLM"),
    "Idefics3ForConditionalGeneration": ("idefics3", "Idefics3ForConditionalGeneration"),
    "SmolVLMForConditionalGeneration": ("idefics3", "SmolVLMForConditionalGeneration"),
    "AyaVisionForConditionalGeneration": ("aya_vision", "AyaVisionForConditionalGeneration"),
    "JinaVLForRanking": ("qwen2_vl", "JinaVLForRanking"),
    "InternVLForConditionalGeneration": ("internvl", "InternVLForConditionalGeneration"),
    "SkyworkR1VChatModel": ("internvl", "InternVLChatModel"),
    "H2OVLChatModel": ("internvl", "InternVLChatModel"),
    "Mistral3ForConditionalGeneration": ("pixtral", "Mistral3ForConditionalGeneration"),
    "PixtralForConditionalGeneration": ("pixtral", "PixtralForConditionalGeneration"),
    "TransformersForCausalLM": ("transformers_backend", "TransformersForCausalLM"),
    "EAGLEModel": ("eagle", "EagleLlamaForCausalLM"),
    "DeepSeekMTPModel": ("mtp", "DeepSeekMTPModel"),
    "MiMoMTPModel": ("mtp", "MiMoMTPModel"),
    "MLPSpeculatorPreTrainedModel": ("mtp", "MLPSpeculatorPreTrainedModel"),
}
_CUSTOM: dict[str, Union[str, type]] = {}


class _Registry:
    def get_supported_archs(self) -> list[str]:
        return sorted(set(_MODELS) | set(_CUSTOM))

    def register_model(self, arch: str, model_cls: Union[str, type]) -> None:
        """``model_cls`` is a class or a ``"module.path:ClassName"`` string."""
        _CUSTOM[arch] = model_cls

    def is_supported(self, arch: str) -> bool:
        return arch in _MODELS or arch in _CUSTOM

    def resolve_model_cls(self, architectures: list[str]) -> tuple[Type, str]:
        for arch in architectures:
            if arch in _CUSTOM:
                c = _CUSTOM[arch]
                if isinstance(c, str):
                    mod, name = c.split(":")
                    c = getattr(importlib.import_module(mod), name)
                return c, arch
            if arch in _MODELS:
                mod, name = _MODELS[arch]
                return getattr(importlib.import_module(f"{__package__}.{mod}"), name), arch
        raise ValueError(f"model architectures {architectures} are not supported; "
                         f"supported: {self.get_supported_archs()}")

    def is_text_generation_model(self, architectures) -> bool:
        return any("CausalLM" in a or "LMHead" in a for a in architectures)


ModelRegistry = _Registry()
