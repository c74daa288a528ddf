"""Encoder-only models: BERT, RoBERTa / XLM-RoBERTa as embedding models and as sequence classifiers
(cross-encoders).  Every prompt runs in one step; its keys and values go through the paged cache
like any prefill and the prefill attention kernel runs with the bidirectional flag, so one kernel
family serves decoders and encoders.
Reference: vllm/model_executor/models/bert.py, roberta.py.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import LayerNorm, VocabParallelEmbedding, get_act_fn
from ..layers.linear import ColumnParallelLinear, QKVParallelLinear, ReplicatedLinear, RowParallelLinear
from ..layers.pooler import Pooler, PoolingType
from ..parallel.state import get_tensor_model_parallel_world_size
from .utils import load_weights_with_mapping

_STACKED = [(".qkv_proj", ".query", "q"), (".qkv_proj", ".key", "k"), (".qkv_proj", ".value", "v")]


class BertLayer(nn.Module):
    """Post-norm block: x = LN(x + attn(x)); x = LN(x + mlp(x))."""

    def __init__(self, config, vllm_config: VllmConfig, prefix: str):
        super().__init__()
        d, H = config.hidden_size, config.num_attention_heads
        tp = get_tensor_model_parallel_world_size()
        D = d // H
        eps = config.layer_norm_eps
        self.qkv_proj = QKVParallelLinear(d, D, H, H, bias=True, prefix=f"{prefix}.attention.self.qkv_proj")
        self.attn = Attention(H // tp, D, D ** -0.5, H // tp, prefix=f"{prefix}.attention.self.attn",
                              cache_dtype=vllm_config.cache_config.cache_dtype, causal=False)
        self.attn_out = RowParallelLinear(d, d, bias=True, prefix=f"{prefix}.attention.output.dense")
        self.attn_norm = LayerNorm(d, eps)
        self.up = ColumnParallelLinear(d, config.intermediate_size, bias=True, prefix=f"{prefix}.intermediate.dense")
        self.down = RowParallelLinear(config.intermediate_size, d, bias=True, prefix=f"{prefix}.output.dense")
        self.out_norm = LayerNorm(d, eps)
        self.act = get_act_fn(config.hidden_act)

    def forward(self, x):
        x = self.attn_norm(x + self.attn_out(self.attn(self.qkv_proj(x))))
        return self.out_norm(x + self.down(self.act(self.up(x))))


_LAYER_RENAMES = ((".attention.self.query.", ".query."), (".attention.self.key.", ".key."),
                  (".attention.self.value.", ".value."), (".attention.output.dense.", ".attn_out."),
                  (".attention.output.LayerNorm.", ".attn_norm."), (".intermediate.dense.", ".up."),
                  (".output.dense.", ".down."), (".output.LayerNorm.", ".out_norm."))


class BertBackbone(nn.Module):
    position_offset = 0          # RoBERTa: padding_idx + 1

    def __init__(self, vllm_config: VllmConfig):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        if getattr(config, "position_embedding_type", "absolute") not in (None, "absolute"):
            raise ValueError("only absolute position embeddings are supported for BERT-style encoders")
        d = config.hidden_size
        self.word_embeddings = VocabParallelEmbedding(config.vocab_size, d)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, d)
        self.token_type_embeddings = nn.Embedding(max(int(getattr(config, "type_vocab_size", 1) or 1), 1), d)
        for e in (self.position_embeddings, self.token_type_embeddings):
            e.weight.requires_grad_(False)
        self.embed_norm = LayerNorm(d, config.layer_norm_eps)
        self.layers = nn.ModuleList([BertLayer(config, vllm_config, f"encoder.layer.{i}")
                                     for i in range(config.num_hidden_layers)])

    def forward(self, input_ids, positions, token_type_ids: Optional[torch.Tensor] = None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.word_embeddings(input_ids)
        x = x + self.position_embeddings(positions + self.position_offset)
        if token_type_ids is None:
            x = x + self.token_type_embeddings.weight[0]
        else:
            x = x + self.token_type_embeddings(token_type_ids)
        x = self.embed_norm(x)
        for layer in self.layers:
            x = layer(x)
        return x


def _backbone_name(name: str) -> Optional[str]:
    """Checkpoint tensor name -> backbone parameter name (None: not part of the backbone)."""
    for p in ("bert.", "roberta.", "model."):
        if name.startswith(p):
            name = name[len(p):]
            break
    if name.startswith("embeddings."):
        name = name[len("embeddings."):].replace("LayerNorm.", "embed_norm.")
        return None if name.startswith("position_ids") else name
    if name.startswith("encoder.layer."):
        name = "layers." + name[len("encoder.layer."):]
        for a, b in _LAYER_RENAMES:
            name = name.replace(a, b)
        return name
    return None


class BertEmbeddingModel(nn.Module):
    """Sentence embeddings: the [CLS] vector (or the pooling the pooler config asks for), L2-normalised."""
    is_pooling_model = True
    is_encoder_only = True
    uses_token_type_ids = True
    supports_pp = False
    backbone_cls = BertBackbone

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        self.config = vllm_config.model_config.hf_config
        self.model = self.backbone_cls(vllm_config)
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.CLS,
                                         normalize=True, softmax=False)

    def get_input_embeddings(self, ids):
        return self.model.word_embeddings(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None, token_type_ids=None):
        return self.model(input_ids, positions, token_type_ids, inputs_embeds)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        return self.pooler(hidden, params)

    def _route(self, name: str) -> Optional[str]:
        b = _backbone_name(name)
        return None if b is None else "model." + b

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        return load_weights_with_mapping(self, weights, _STACKED, rename=self._route)


class RobertaBackbone(BertBackbone):
    def __init__(self, vllm_config: VllmConfig):
        super().__init__(vllm_config)
        self.position_offset = int(getattr(self.config, "pad_token_id", 1) or 0) + 1


class RobertaEmbeddingModel(BertEmbeddingModel):
    backbone_cls = RobertaBackbone
    uses_token_type_ids = False


class BertForSequenceClassification(BertEmbeddingModel):
    """[CLS] -> dense + tanh (the BERT pooler) -> classifier.  One label: a cross-encoder score."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        d = self.config.hidden_size
        labels = getattr(self.config, "id2label", None)
        n = len(labels) if labels else int(getattr(self.config, "num_labels", 2) or 2)
        self.pooler_dense = ReplicatedLinear(d, d, bias=True, prefix="bert.pooler.dense")
        self.classifier = ReplicatedLinear(d, n, bias=True, prefix="classifier")
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.CLS,
                                         normalize=False, softmax=True)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        x = self.pooler.extract(hidden).unsqueeze(0)
        return self.pooler.head(self.classifier(torch.tanh(self.pooler_dense(x))).squeeze(0), params)

    def _route(self, name: str) -> Optional[str]:
        if name.startswith("classifier."):
            return name
        if name.startswith("bert.pooler.dense."):
            return "pooler_dense." + name[len("bert.pooler.dense."):]
        return super()._route(name)


class RobertaForSequenceClassification(RobertaEmbeddingModel):
    """<s> -> dense + tanh -> out_proj (the RoBERTa classification head)."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        d = self.config.hidden_size
        labels = getattr(self.config, "id2label", None)
        n = len(labels) if labels else int(getattr(self.config, "num_labels", 2) or 2)
        self.head_dense = ReplicatedLinear(d, d, bias=True, prefix="classifier.dense")
        self.head_out = ReplicatedLinear(d, n, bias=True, prefix="classifier.out_proj")
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.CLS,
                                         normalize=False, softmax=True)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        x = self.pooler.extract(hidden).unsqueeze(0)
        return self.pooler.head(self.head_out(torch.tanh(self.head_dense(x))).squeeze(0), params)

    def _route(self, name: str) -> Optional[str]:
        if name.startswith("classifier.dense."):
            return "head_dense." + name[len("classifier.dense."):]
        if name.startswith("classifier.out_proj."):
            return "head_out." + name[len("classifier.out_proj."):]
        return super()._route(name)
