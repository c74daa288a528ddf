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
