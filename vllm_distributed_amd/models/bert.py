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


# ---- rotary encoders (Nomic BERT) -----------------------------------------------------------------------------
class NomicBertLayer(nn.Module):
    """Post-norm block with rotary positions and a gated MLP, no biases:
    x = LN(x + o(attn(rope(q, k), v)));  x = LN(x + down(act(gate(x)) * up(x)))."""

    linear_bias = False
    act_name: Optional[str] = None

    def __init__(self, config, vllm_config: VllmConfig, prefix: str, rotary):
        super().__init__()
        from ..layers.basic import get_act_and_mul_fn
        from ..layers.linear import MergedColumnParallelLinear
        d, H = config.hidden_size, config.num_attention_heads
        D = getattr(config, "head_dim", None) or d // H
        tp = get_tensor_model_parallel_world_size()
        eps = config.layer_norm_eps
        self.rotary = rotary
        b = self.linear_bias          # every projection but the fused gate / up one
        self.qkv_proj = QKVParallelLinear(d, D, H, H, bias=b, prefix=f"{prefix}.self_attn.qkv_proj")
        self.attn = Attention(H // tp, D, D ** -0.5, H // tp, prefix=f"{prefix}.self_attn.attn",
                              cache_dtype=vllm_config.cache_config.cache_dtype, causal=False)
        self.o_proj = RowParallelLinear(H * D, d, bias=b, prefix=f"{prefix}.self_attn.o_proj")
        self.post_attention_layernorm = LayerNorm(d, eps)
        self.gate_up_proj = MergedColumnParallelLinear(d, [config.intermediate_size] * 2, bias=False,
                                                       prefix=f"{prefix}.mlp.gate_up_proj")
        self.down_proj = RowParallelLinear(config.intermediate_size, d, bias=b, prefix=f"{prefix}.mlp.down_proj")
        self.post_mlp_layernorm = LayerNorm(d, eps)
        self.act = get_act_and_mul_fn(self.act_name or getattr(config, "hidden_act", "silu"))

    def forward(self, x, positions):
        x = self.post_attention_layernorm(x + self.o_proj(self.attn(self.qkv_proj(x), positions, self.rotary)))
        return self.post_mlp_layernorm(x + self.down_proj(self.act(self.gate_up_proj(x))))


class NomicBertBackbone(nn.Module):
    layer_cls = NomicBertLayer

    def __init__(self, vllm_config: VllmConfig):
        super().__init__()
        from ..layers.rotary import get_rope
        config = vllm_config.model_config.hf_config
        self.config = config
        d = config.hidden_size
        D = getattr(config, "head_dim", None) or d // config.num_attention_heads
        rp = getattr(config, "rope_parameters", None) or {}
        theta = float(getattr(config, "rope_theta", None) or getattr(config, "rotary_emb_base", None)
                      or rp.get("rope_theta", 10000.0))
        scaling = getattr(config, "rope_scaling", None) if self.layer_cls is not NomicBertLayer else None
        rotary = get_rope(D, D, config.max_position_embeddings, theta, True, dict(scaling) if scaling else None)
        self.word_embeddings = VocabParallelEmbedding(config.vocab_size, d)
        self.token_type_embeddings = nn.Embedding(max(int(getattr(config, "type_vocab_size", 1) or 1), 1), d)
        self.token_type_embeddings.weight.requires_grad_(False)
        self.embed_norm = LayerNorm(d, config.layer_norm_eps)
        self.layers = nn.ModuleList([self.layer_cls(config, vllm_config, f"encoder.layers.{i}", rotary)
                                     for i in range(config.num_hidden_layers)])

    def forward(self, input_ids, positions, token_type_ids: Optional[torch.Tensor] = None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.word_embeddings(input_ids)
        x = x + (self.token_type_embeddings.weight[0] if token_type_ids is None
                 else self.token_type_embeddings(token_type_ids))
        x = self.embed_norm(x)
        for layer in self.layers:
            x = layer(x, positions)
        return x


class NomicBertModel(BertEmbeddingModel):
    """Sentence embeddings from a rotary BERT (mean pooling, L2-normalised), in the layout transformers writes:
    ``embeddings.{word,token_type}_embeddings``, ``encoder.layers.N.self_attn.{q,k,v,o}_proj``,
    ``mlp.{gate,up,down}_proj``, ``post_{attention,mlp}_layernorm``.
    Reference: vllm/model_executor/models/bert_with_rope.py (NomicBertModel :520-600)."""
    backbone_cls = NomicBertBackbone
    uses_token_type_ids = True
    _NB_STACKED = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
                   (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.MEAN,
                                         normalize=True, softmax=False)

    # original (nomic-ai) names -> the module tree above; transformers writes these names too when it saves
    _ORIG = (("emb_ln.", "embeddings.LayerNorm."), (".attn.out_proj.", ".self_attn.o_proj."), (".mlp.fc11.", ".mlp.up_proj."),
             (".mlp.fc12.", ".mlp.gate_proj."), (".mlp.fc2.", ".mlp.down_proj."), (".norm1.", ".post_attention_layernorm."),
             (".norm2.", ".post_mlp_layernorm."))

    def _route(self, name: str) -> Optional[str]:
        for p in ("nomic_bert.", "bert.", "model."):
            if name.startswith(p):
                name = name[len(p):]
                break
        for a, b in self._ORIG:
            name = name.replace(a, b)
        if name.startswith("layers."):
            name = "encoder." + name
        if name.startswith("embeddings."):
            name = name[len("embeddings."):].replace("LayerNorm.", "embed_norm.")
            return None if name.startswith(("position_ids", "token_type_ids")) else "model." + name
        if name.startswith("encoder.layers."):
            name = "layers." + name[len("encoder.layers."):]
            for a, b in ((".self_attn.q_proj.", ".q_proj."), (".self_attn.k_proj.", ".k_proj."),
                         (".self_attn.v_proj.", ".v_proj."), (".self_attn.o_proj.", ".o_proj."),
                         (".mlp.gate_proj.", ".gate_proj."), (".mlp.up_proj.", ".up_proj."),
                         (".mlp.down_proj.", ".down_proj."), (".attn.Wqkv.", ".Wqkv.")):
                name = name.replace(a, b)
            return "model." + name
        return None                                     # pooler / prediction heads are not used for embeddings

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def split_fused():
            for n, t in weights:
                if ".attn.Wqkv." in n:                  # fused [q; k; v] rows -> the three logical projections
                    for part, piece in zip(("q_proj", "k_proj", "v_proj"), t.chunk(3, dim=0)):
                        yield n.replace(".attn.Wqkv.", f".self_attn.{part}."), piece
                else:
                    yield n, t
        return load_weights_with_mapping(self, split_fused(), self._NB_STACKED, rename=self._route)


class GteNewLayer(NomicBertLayer):
    linear_bias = True
    act_name = "gelu"               # gelu(gate) * up


class GteNewBackbone(NomicBertBackbone):
    layer_cls = GteNewLayer


class GteNewModel(NomicBertModel):
    """Alibaba-NLP "new" encoder (gte-multilingual-base ...): rotary BERT with biased projections, a bias-free fused
    ``up_gate_proj`` (up rows first), GELU gate, optional fixed-NTK rope scaling, CLS pooling.
    Checkpoint names: ``new.embeddings.*``, ``encoder.layer.N.attention.{qkv_proj,o_proj}``, ``mlp.{up_gate_proj,
    down_proj}``, ``attn_ln``, ``mlp_ln``.  Reference: vllm/model_executor/models/bert_with_rope.py:487-540,
    vllm/model_executor/models/config.py:21-40."""
    backbone_cls = GteNewBackbone

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.CLS,
                                         normalize=True, softmax=False)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def rename():
            for n, t in weights:
                if n.startswith("new."):
                    n = n[4:]
                if n.startswith(("classifier", "pooler")):
                    continue
                n = n.replace("encoder.layer.", "encoder.layers.").replace(".attn_ln.", ".post_attention_layernorm.") \
                     .replace(".mlp_ln.", ".post_mlp_layernorm.").replace(".attention.o_proj.", ".self_attn.o_proj.")
                if ".attention.qkv_proj." in n:
                    for part, piece in zip(("q_proj", "k_proj", "v_proj"), t.chunk(3, dim=0)):
                        yield n.replace(".attention.qkv_proj.", f".self_attn.{part}."), piece
                elif ".mlp.up_gate_proj." in n:
                    up, gate = t.chunk(2, dim=0)
                    yield n.replace(".mlp.up_gate_proj.", ".mlp.up_proj."), up
                    yield n.replace(".mlp.up_gate_proj.", ".mlp.gate_proj."), gate
                else:
                    yield n, t
        return load_weights_with_mapping(self, rename(), self._NB_STACKED, rename=self._route)


class SnowflakeGteNewModel(GteNewModel):
    """snowflake-arctic-embed-m-v2.0: the same encoder saved without the ``new.`` level (architecture ``GteModel``)."""


# ---- ModernBERT ---------------------------------------------------------------------------------------------------
class ModernBertLayer(nn.Module):
    """Pre-norm block (no norm in front of the first attention), fused Wqkv, GeGLU MLP with a fused ``Wi``
    (activation half first).  Global layers attend over the whole sequence through the paged kernels; local layers
    see |i - j| <= local_attention / 2 and compute that band directly on the step's tokens (an encoder runs every
    prompt in one step, so nothing has to be cached for them)."""

    def __init__(self, config, vllm_config: VllmConfig, idx: int, is_global: bool, rotary):
        super().__init__()
        from ..layers.basic import get_act_and_mul_fn
        from ..layers.linear import MergedColumnParallelLinear
        d, H = config.hidden_size, config.num_attention_heads
        if get_tensor_model_parallel_world_size() != 1:
            raise ValueError("ModernBERT runs on one rank per replica (use data parallelism to scale it)")
        self.H, self.D = H, d // H
        eps, nb = float(getattr(config, "norm_eps", 1e-5)), bool(getattr(config, "norm_bias", False))
        ab, mb = bool(getattr(config, "attention_bias", False)), bool(getattr(config, "mlp_bias", False))
        self.attn_norm = nn.LayerNorm(d, eps=eps, bias=nb) if idx > 0 else nn.Identity()
        self.qkv_proj = QKVParallelLinear(d, self.D, H, H, bias=ab, prefix=f"layers.{idx}.attn.Wqkv")
        self.o_proj = RowParallelLinear(d, d, bias=ab, prefix=f"layers.{idx}.attn.Wo")
        self.is_global, self.rotary = is_global, rotary
        self.half_window = int(getattr(config, "local_attention", 128)) // 2
        if is_global:
            self.attn = Attention(H, self.D, self.D ** -0.5, H, prefix=f"layers.{idx}.attn.attn",
                                  cache_dtype=vllm_config.cache_config.cache_dtype, causal=False)
        self.mlp_norm = nn.LayerNorm(d, eps=eps, bias=nb)
        self.gate_up_proj = MergedColumnParallelLinear(d, [config.intermediate_size] * 2, bias=mb,
                                                       prefix=f"layers.{idx}.mlp.Wi")
        self.down_proj = RowParallelLinear(config.intermediate_size, d, bias=mb, prefix=f"layers.{idx}.mlp.Wo")
        self.act = get_act_and_mul_fn(getattr(config, "hidden_activation", "gelu"))

    def _local(self, qkv: torch.Tensor, positions: torch.Tensor, band: Optional[torch.Tensor]) -> torch.Tensor:
        T, d = qkv.shape[0], self.H * self.D
        q, k = qkv[:, :d].contiguous(), qkv[:, d:2 * d].contiguous()
        self.rotary(positions[:T], q, k)
        q, k, v = (t.reshape(T, self.H, self.D).transpose(0, 1)[None] for t in (q, k, qkv[:, 2 * d:]))
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=band)
        return o[0].transpose(0, 1).reshape(T, d)

    def forward(self, x, positions, band):
        qkv = self.qkv_proj(self.attn_norm(x))
        a = self.attn(qkv, positions, self.rotary) if self.is_global else self._local(qkv, positions, band)
        x = x + self.o_proj(a)
        return x + self.down_proj(self.act(self.gate_up_proj(self.mlp_norm(x))))


class ModernBertBackbone(nn.Module):
    def __init__(self, vllm_config: VllmConfig):
        super().__init__()
        from ..layers.rotary import get_rope
        config = vllm_config.model_config.hf_config
        self.config = config
        d = config.hidden_size
        D = d // config.num_attention_heads
        eps, nb = float(getattr(config, "norm_eps", 1e-5)), bool(getattr(config, "norm_bias", False))
        rp = getattr(config, "rope_parameters", None) or {}
        g_theta = float((rp.get("full_attention") or {}).get("rope_theta") or getattr(config, "global_rope_theta", 160000.0))
        l_theta = float((rp.get("sliding_attention") or {}).get("rope_theta")
                        or getattr(config, "local_rope_theta", None) or g_theta)
        kinds = list(getattr(config, "layer_types", None) or [])
        every = int(getattr(config, "global_attn_every_n_layers", 3) or 1)
        self.word_embeddings = VocabParallelEmbedding(config.vocab_size, d)
        self.embed_norm = nn.LayerNorm(d, eps=eps, bias=nb)
        ropes = {True: get_rope(D, D, config.max_position_embeddings, g_theta, True),
                 False: get_rope(D, D, config.max_position_embeddings, l_theta, True)}
        self.layers = nn.ModuleList()
        for i in range(config.num_hidden_layers):
            is_global = (kinds[i] == "full_attention") if kinds else (i % every == 0)
            self.layers.append(ModernBertLayer(config, vllm_config, i, is_global, ropes[is_global]))
        self.final_norm = nn.LayerNorm(d, eps=eps, bias=nb)
        self.half_window = int(getattr(config, "local_attention", 128)) // 2

    def forward(self, input_ids, positions, token_type_ids: Optional[torch.Tensor] = None, inputs_embeds=None):
        from ..forward_context import get_forward_context
        x = self.embed_norm(inputs_embeds if inputs_embeds is not None else self.word_embeddings(input_ids))
        T = x.shape[0]
        md = get_forward_context().attn_metadata
        pos = positions[:T]
        near = (pos[:, None] - pos[None, :]).abs() <= self.half_window
        if md is not None:                               # same sequence: tokens between the same two start offsets
            S = md.num_decode_seqs + md.num_prefill_seqs
            seq = torch.searchsorted(md.query_start_loc[1:S + 1].to(torch.int64),
                                     torch.arange(T, device=x.device), right=True)
            near &= seq[:, None] == seq[None, :]
        for layer in self.layers:
            x = layer(x, positions, near)
        return self.final_norm(x)


class ModernBertModel(BertEmbeddingModel):
    """ModernBERT sentence embeddings (CLS by default).  Checkpoint names: ``model.embeddings.{tok_embeddings,norm}``,
    ``model.layers.N.{attn_norm,attn.Wqkv,attn.Wo,mlp_norm,mlp.Wi,mlp.Wo}``, ``model.final_norm``.
    Reference: vllm/model_executor/models/modernbert.py:40-339."""
    backbone_cls = ModernBertBackbone
    uses_token_type_ids = False
    _RENAMES = (("embeddings.tok_embeddings.", "word_embeddings."), ("embeddings.norm.", "embed_norm."),
                (".attn.Wo.", ".o_proj."), (".mlp.Wo.", ".down_proj."))

    def _route(self, name: str) -> Optional[str]:
        if name.startswith("model."):
            name = name[6:]
        if name.startswith(("head.", "classifier.", "decoder.")):
            return None
        for a, b in self._RENAMES:
            name = name.replace(a, b)
        return "model." + name

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def split():
            for n, t in weights:
                if ".attn.Wqkv." in n:
                    for part, piece in zip(("q", "k", "v"), t.chunk(3, dim=0)):
                        yield n.replace(".attn.Wqkv.", f".{part}_part."), piece
                elif ".mlp.Wi." in n:                   # activation half first, then the linear half
                    a, g = t.chunk(2, dim=0)
                    yield n.replace(".mlp.Wi.", ".act_part."), a
                    yield n.replace(".mlp.Wi.", ".lin_part."), g
                else:
                    yield n, t
        stacked = [(".qkv_proj", ".q_part", "q"), (".qkv_proj", ".k_part", "k"), (".qkv_proj", ".v_part", "v"),
                   (".gate_up_proj", ".act_part", 0), (".gate_up_proj", ".lin_part", 1)]
        return load_weights_with_mapping(self, split(), stacked, rename=self._route)


class ModernBertForSequenceClassification(ModernBertModel):
    """CLS (or mean) -> dense + activation + LayerNorm -> classifier."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        c = self.config
        d = c.hidden_size
        labels = getattr(c, "id2label", None)
        n = len(labels) if labels else int(getattr(c, "num_labels", 2) or 2)
        self.head_dense = nn.Linear(d, d, bias=bool(getattr(c, "classifier_bias", False)))
        self.head_norm = nn.LayerNorm(d, eps=float(getattr(c, "norm_eps", 1e-5)), bias=bool(getattr(c, "norm_bias", False)))
        self.head_act = get_act_fn(getattr(c, "classifier_activation", "gelu"))
        self.classifier = nn.Linear(d, n)
        kind = PoolingType.MEAN if getattr(c, "classifier_pooling", "cls") == "mean" else PoolingType.CLS
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), kind,
                                         normalize=False, softmax=True)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        x = self.pooler.extract(hidden)
        return self.pooler.head(self.classifier(self.head_norm(self.head_act(self.head_dense(x)))), params)

    def _route(self, name: str) -> Optional[str]:
        if name.startswith("classifier."):
            return name
        if name.startswith("head.dense."):
            return "head_dense." + name[len("head.dense."):]
        if name.startswith("head.norm."):
            return "head_norm." + name[len("head.norm."):]
        return super()._route(name)
