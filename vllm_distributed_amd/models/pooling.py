"""Pooling variants of the Llama-family decoder: embeddings (last-token, L2-normalised) and
sequence classification / scoring (linear ``score`` head on the last token).
Reference: vllm/model_executor/models/adapters.py (as_embedding_model, as_seq_cls_model),
vllm/model_executor/models/llama.py (LlamaModel as embedding backbone)."""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.linear import ReplicatedLinear
from ..layers.pooler import Pooler, PoolingType
from .llama import LlamaModel
from .utils import load_weights_with_mapping, maybe_prefix

_STACKED = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
            (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]


class LlamaEmbeddingModel(nn.Module):
    is_pooling_model = True
    supports_pp = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        self.config = vllm_config.model_config.hf_config
        self.model = LlamaModel(vllm_config, prefix=maybe_prefix(prefix, "model"))
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.LAST,
                                         normalize=True, softmax=False)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        return self.pooler(hidden, params)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def rename(n: str):
            if n.startswith("lm_head."):
                return None
            return n if n.startswith("model.") else "model." + n      # bare backbones (sentence-transformers style)
        return load_weights_with_mapping(self, weights, _STACKED, rename=rename)


class LlamaForSequenceClassification(LlamaEmbeddingModel):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        labels = getattr(self.config, "id2label", None)      # HF stores the label map, not num_labels
        n = len(labels) if labels else int(getattr(self.config, "num_labels", 2) or 2)
        self.score = ReplicatedLinear(self.config.hidden_size, n, bias=False, prefix=maybe_prefix(prefix, "score"))
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.LAST,
                                         normalize=False, softmax=True)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        x = self.pooler.extract(hidden)
        return self.pooler.head(self.score(x.unsqueeze(0) if x.dim() == 1 else x).squeeze(0), params)

    def load_weights(self, weights):
        def rename(n: str):
            if n.startswith("lm_head."):
                return None
            return n if n.startswith(("model.", "score.")) else "model." + n
        return load_weights_with_mapping(self, weights, _STACKED, rename=rename)


class Qwen2ForRewardModel(LlamaEmbeddingModel):
    """Per-token rewards: Qwen2 backbone, then Linear -> ReLU -> Linear to ``num_labels`` values for every
    prompt token (pooling type ALL).  The process-reward variant has two labels and a softmax.
    Reference: vllm/model_executor/models/qwen2_rm.py.  transformers does not carry these classes, so the
    test checks the head against a plain PyTorch forward on the backbone's own hidden states."""
    num_labels = 1
    softmax = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        d = self.config.hidden_size
        self.score_in = ReplicatedLinear(d, d, bias=True, prefix=maybe_prefix(prefix, "score.0"))
        self.score_out = ReplicatedLinear(d, self.num_labels, bias=True, prefix=maybe_prefix(prefix, "score.2"))
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.ALL,
                                         normalize=False, softmax=self.softmax)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        x = self.pooler.extract(hidden)
        x = x.unsqueeze(0) if x.dim() == 1 else x
        return self.pooler.head(self.score_out(torch.relu(self.score_in(x))), params)

    def load_weights(self, weights):
        def rename(n: str):
            if n.startswith("lm_head."):
                return None
            if n.startswith("score.0."):
                return "score_in." + n[len("score.0."):]
            if n.startswith("score.2."):
                return "score_out." + n[len("score.2."):]
            return n if n.startswith("model.") else "model." + n
        return load_weights_with_mapping(self, weights, _STACKED, rename=rename)


class Qwen2ForProcessRewardModel(Qwen2ForRewardModel):
    num_labels = 2
    softmax = True


# ---- adapters over any decoder of the zoo ---------------------------------------------------------------
def _resolve(module: str, name: str):
    import importlib
    return getattr(importlib.import_module(f"{__package__}.{module}"), name)


class _DecoderAdapter(nn.Module):
    """A causal LM class of this package used as a pooling backbone: its forward already returns the final
    hidden states; the output head is built but never loaded or used.
    Reference: vllm/model_executor/models/adapters.py (as_embedding_model :45-190, as_seq_cls_model :193-320)."""
    is_pooling_model = True
    supports_pp = False
    base: tuple[str, str] = ("llama", "LlamaForCausalLM")
    head_names: tuple[str, ...] = ()           # checkpoint prefixes of this adapter's own head

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        self.config = vllm_config.model_config.hf_config
        self.inner = _resolve(*self.base)(vllm_config, prefix)
        self.hidden_size = vllm_config.model_config.get_hidden_size()
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.LAST,
                                         normalize=True, softmax=False)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.inner(input_ids, positions, intermediate_tensors, inputs_embeds)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        return self.pooler(hidden, params)

    def _own_weight(self, name: str, tensor: torch.Tensor) -> bool:
        return False

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        backbone_prefixes = ("model.", "transformer.", "backbone.", "lm_head.", "gpt_neox.")

        own: set[str] = set()

        def rest():
            for n, t in weights:
                if self.head_names and n.startswith(self.head_names):
                    if self._own_weight(n, t):
                        own.add(n)
                elif n.startswith("lm_head."):
                    continue
                else:
                    # bare backbones (AutoModel checkpoints) carry no "model." prefix
                    yield (n if n.startswith(backbone_prefixes) else "model." + n), t
        loaded = {"inner." + n for n in (self.inner.load_weights(rest()) or ())}
        # the unused output head keeps its initial values
        loaded |= {n for n, _ in self.named_parameters() if n.startswith("inner.lm_head.")}
        return loaded | own


class _SeqClsAdapter(_DecoderAdapter):
    head_names = ("score.",)

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        labels = getattr(self.config, "id2label", None)
        n = len(labels) if labels else int(getattr(self.config, "num_labels", 2) or 2)
        self.score = ReplicatedLinear(self.hidden_size, n, bias=False, prefix=maybe_prefix(prefix, "score"))
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.LAST,
                                         normalize=False, softmax=True)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        x = self.pooler.extract(hidden)
        return self.pooler.head(self.score(x.unsqueeze(0) if x.dim() == 1 else x).squeeze(0), params)

    def _own_weight(self, name, tensor):
        if name == "score.weight":
            self.score.weight.data.copy_(tensor)
        return True


class GPT2ForSequenceClassification(_SeqClsAdapter):
    base = ("opt", "GPT2LMHeadModel")


class GemmaForSequenceClassification(_SeqClsAdapter):
    base = ("llama_family", "GemmaForCausalLM")


class JambaForSequenceClassification(_SeqClsAdapter):
    base = ("jamba", "JambaForCausalLM")


class Gemma2EmbeddingModel(_DecoderAdapter):
    base = ("llama_family", "Gemma2ForCausalLM")


class InternLM2ForRewardModel(_DecoderAdapter):
    """InternLM2 backbone with a one-value head on every token (pooling type ALL).
    Reference: vllm/model_executor/models/internlm2.py (InternLM2ForRewardModel)."""
    base = ("llama_variants", "InternLM2ForCausalLM")
    head_names = ("v_head.",)

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        self.v_head = ReplicatedLinear(self.hidden_size, 1, bias=False, prefix=maybe_prefix(prefix, "v_head"))
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.ALL,
                                         normalize=False, softmax=False)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        x = self.pooler.extract(hidden)
        x = x.unsqueeze(0) if x.dim() == 1 else x
        return self.pooler.head(self.v_head(x), params)

    def _own_weight(self, name, tensor):
        if name == "v_head.weight":
            self.v_head.weight.data.copy_(tensor)
        return True
