"""Draft networks for speculative decoding: EAGLE (one Llama layer over [embedding ; target
hidden]) and Medusa (residual-MLP heads).
Reference: vllm/model_executor/models/llama_eagle.py, medusa.py."""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.basic import LogitsProcessor, ParallelLMHead, VocabParallelEmbedding
from ..layers.linear import ReplicatedLinear
from .llama import LlamaAttention, LlamaDecoderLayer
from .utils import load_weights_with_mapping, maybe_prefix


class _EagleLayer(LlamaDecoderLayer):
    """First (only) draft layer: EAGLE-1 checkpoints have no input norm on it."""

    def __init__(self, config, vllm_config, prefix: str, no_input_norm: bool):
        super().__init__(config, vllm_config, None, prefix)
        if no_input_norm:
            del self.input_layernorm
            self.input_layernorm = None

    def forward(self, positions, hidden_states, residual):
        if self.input_layernorm is None:
            residual = hidden_states
        elif residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class EagleLlamaForCausalLM(nn.Module):
    share_embeddings = True

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        H = config.hidden_size
        n = int(getattr(config, "num_hidden_layers", 1) or 1)
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, H, prefix="draft.embed_tokens")
        self.fc = ReplicatedLinear(2 * H, H, bias=bool(getattr(config, "eagle_fc_bias", False)), prefix="draft.fc")
        # layer indices continue after the target's so that KV-cache layer names stay unique
        self.layers = nn.ModuleList([_EagleLayer(config, vllm_config, f"draft.layers.{i}", no_input_norm=(i == 0))
                                     for i in range(n)])
        self.lm_head = ParallelLMHead(config.vocab_size, H, prefix="draft.lm_head")
        self.logits_processor = LogitsProcessor(config.vocab_size, config.vocab_size)
        self._own_embed = self._own_head = False

    def set_shared(self, embed, lm_head, logits_processor) -> None:
        if not self._own_embed:
            self.embed_tokens = embed
        if not self._own_head:
            self.lm_head, self.logits_processor = lm_head, logits_processor

    def forward(self, input_ids, positions, hidden_states):
        x = self.fc(torch.cat((self.embed_tokens(input_ids), hidden_states), dim=-1))
        residual = None
        for layer in self.layers:
            x, residual = layer(positions, x, residual)
        return x + residual

    def compute_logits(self, hidden_states):
        return self.logits_processor(self.lm_head, hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def rename(name: str):
            if name.startswith("model."):
                name = name[len("model."):]
            if name.startswith("embed_tokens."):
                self._own_embed = True
            if name.startswith("lm_head."):
                self._own_head = True
            return name
        stacked = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
                   (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]
        load_weights_with_mapping(self, weights, stacked, rename=rename)
        return None      # embedding / head may legitimately be absent (shared with the target)


class _EagleMiniCPMLayer(LlamaDecoderLayer):
    """MiniCPM draft block: both branches enter the residual stream scaled by scale_depth / sqrt(mup_denominator)."""

    def __init__(self, config, vllm_config, prefix: str):
        super().__init__(config, vllm_config, None, prefix)
        self.branch_scale = float(getattr(config, "scale_depth", 1.0)) / \
            float(getattr(config, "mup_denominator", None) or config.num_hidden_layers) ** 0.5

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states) * self.branch_scale
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states) * self.branch_scale, residual


class EagleMiniCPMForCausalLM(EagleLlamaForCausalLM):
    """EAGLE draft of the MiniCPM family: the token embedding (times ``scale_emb``) and the target's hidden state
    are each RMS-normalised before the fusing projection, the blocks scale their branches like MiniCPM's, and the
    output is divided by hidden_size / dim_model_base before the head.  Checkpoint names: ``eagle_layers.N``,
    ``input_norm1`` / ``input_norm2``.
    Reference: vllm/model_executor/models/minicpm_eagle.py (EagleMiniCPMModel :136-230, EagleMiniCPMForCausalLM
    :288-388)."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        from ..layers.basic import RMSNorm
        config = vllm_config.model_config.hf_config
        self.config = config
        H = config.hidden_size
        n = int(getattr(config, "num_hidden_layers", 1) or 1)
        eps = float(getattr(config, "rms_norm_eps", 1e-5))
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, H, prefix="draft.embed_tokens")
        self.fc = ReplicatedLinear(2 * H, H, bias=False, prefix="draft.fc")
        self.input_norm1, self.input_norm2 = RMSNorm(H, eps), RMSNorm(H, eps)
        self.layers = nn.ModuleList([_EagleMiniCPMLayer(config, vllm_config, f"draft.layers.{i}") for i in range(n)])
        self.lm_head = ParallelLMHead(config.vocab_size, H, prefix="draft.lm_head")
        self.logits_processor = LogitsProcessor(config.vocab_size, config.vocab_size)
        self.scale_emb = float(getattr(config, "scale_emb", 1.0) or 1.0)
        base = float(getattr(config, "dim_model_base", 0) or 0)
        self.scale_width = H / base if base else 1.0
        self._own_embed = self._own_head = False

    def forward(self, input_ids, positions, hidden_states):
        e = self.input_norm1(self.embed_tokens(input_ids) * self.scale_emb)
        x = self.fc(torch.cat((e, self.input_norm2(hidden_states)), dim=-1))
        residual = None
        for layer in self.layers:
            x, residual = layer(positions, x, residual)
        return (x + residual) / self.scale_width

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def renamed():
            for name, t in weights:
                yield name.replace("eagle_layers.", "layers."), t
        out = super().load_weights(renamed())
        if getattr(self.config, "tie_word_embeddings", False) and self._own_embed:
            self.lm_head.tie_weights(self.embed_tokens)
            self._own_head = True
        return out


class _Eagle3Attention(LlamaAttention):
    """Queries, keys and values are projected from [normed token embedding | normed hidden state] (2 H wide)."""

    def _make_projections(self, hidden_size, num_heads, num_kv_heads, bias, o_bias, quant_config, prefix):
        from ..layers.linear import QKVParallelLinear, RowParallelLinear
        self.qkv_proj = QKVParallelLinear(2 * hidden_size, self.head_dim, num_heads, num_kv_heads, bias=bias,
                                          quant_config=quant_config, prefix=f"{prefix}.qkv_proj")
        self.o_proj = RowParallelLinear(num_heads * self.head_dim, hidden_size, bias=o_bias,
                                        quant_config=quant_config, prefix=f"{prefix}.o_proj")


class Eagle3LlamaForCausalLM(nn.Module):
    """EAGLE-3 draft: the target's residual stream at three depths (low / middle / high), concatenated and
    projected to H, and the embedding of the next token enter ONE decoder layer; later draft steps feed the layer's
    own un-normalised output back instead.  The head predicts over a reduced vocabulary, ``d2t`` maps its ids back.
    Reference: vllm/model_executor/models/llama_eagle3.py:28-263, vllm/v1/spec_decode/eagle.py."""
    share_embeddings = True
    is_eagle3 = True

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        from ..layers.basic import RMSNorm
        from .llama import LlamaMLP
        config = vllm_config.model_config.hf_config
        self.config = config
        H = config.hidden_size
        Ht = int(getattr(config, "target_hidden_size", None) or H)
        eps = float(getattr(config, "rms_norm_eps", 1e-6))
        self.vocab = int(config.vocab_size)
        self.draft_vocab = int(getattr(config, "draft_vocab_size", None) or self.vocab)
        self.embed_tokens = VocabParallelEmbedding(self.vocab, H, prefix="draft.embed_tokens")
        self.fc = ReplicatedLinear(3 * Ht, H, bias=False, prefix="draft.fc")
        rp = getattr(config, "rope_parameters", None) or {}
        scaling = getattr(config, "rope_scaling", None)
        theta = float(getattr(config, "rope_theta", None) or rp.get("rope_theta") or 10000.0)
        layer = self.midlayer = nn.Module()
        layer.self_attn = _Eagle3Attention(config, H, config.num_attention_heads,
                                           getattr(config, "num_key_value_heads", None) or config.num_attention_heads,
                                           theta, scaling if scaling and ("rope_type" in scaling or "type" in scaling) else None,
                                           vllm_config.model_config.max_model_len,
                                           bias=bool(getattr(config, "attention_bias", False)),
                                           prefix="draft.layers.0.self_attn",
                                           cache_dtype=vllm_config.cache_config.cache_dtype)
        layer.mlp = LlamaMLP(H, config.intermediate_size, bool(getattr(config, "mlp_bias", False)), None,
                             "draft.layers.0.mlp")
        layer.input_layernorm, layer.hidden_norm = RMSNorm(H, eps), RMSNorm(H, eps)
        layer.post_attention_layernorm = RMSNorm(H, eps)
        self.norm = RMSNorm(H, eps)
        self.lm_head = ParallelLMHead(self.draft_vocab, H, prefix="draft.lm_head")
        self.logits_processor = LogitsProcessor(self.draft_vocab, self.draft_vocab)
        self.register_buffer("d2t", torch.zeros(self.draft_vocab, dtype=torch.int64), persistent=False)
        self._own_embed = False

    def set_shared(self, embed, lm_head, logits_processor) -> None:
        if not self._own_embed:
            self.embed_tokens = embed               # the head is always the draft's own

    def forward(self, input_ids, positions, hidden_states):
        m = self.midlayer
        e = self.embed_tokens(input_ids)
        if hidden_states.shape[-1] != e.shape[-1]:
            hidden_states = self.fc(hidden_states)
        x = torch.cat((m.input_layernorm(e), m.hidden_norm(hidden_states)), dim=-1)
        x, residual = m.post_attention_layernorm(m.self_attn.o_proj(m.self_attn.attn(
            m.self_attn.qkv_proj(x), positions, m.self_attn.rotary_emb)), hidden_states)
        return m.mlp(x) + residual

    def compute_logits(self, hidden_states):
        lg = self.logits_processor(self.lm_head, self.norm(hidden_states))
        if self.draft_vocab == self.vocab:
            return lg
        full = torch.full((lg.shape[0], self.vocab), float("-inf"), dtype=lg.dtype, device=lg.device)
        full[:, torch.arange(self.draft_vocab, device=lg.device) + self.d2t] = lg
        return full

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def rename(name: str):
            if name.startswith("model."):
                name = name[len("model."):]
            if name.startswith("layers.0."):
                name = "midlayer." + name[len("layers.0."):]
            if name.startswith("embed_tokens."):
                self._own_embed = True
            return name
        rest = []
        for n, t in weights:
            if n.endswith("d2t"):
                self.d2t.copy_(t.to(torch.int64))
            elif not n.endswith("t2d"):
                rest.append((n, t))
        stacked = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
                   (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]
        load_weights_with_mapping(self, rest, stacked, rename=rename)
        return None


class _ResidualBlock(nn.Module):
    def __init__(self, H: int, n: int):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(H, H) for _ in range(n)])

    def forward(self, x):
        for l in self.layers:
            x = x + torch.nn.functional.silu(l(x))
        return x


class MedusaModel(nn.Module):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        H = config.hidden_size
        n_heads = int(getattr(config, "num_heads", None) or getattr(config, "medusa_num_heads", 1))
        n_layers = int(getattr(config, "num_hidden_layers", None) or getattr(config, "medusa_num_layers", 1))
        V = int(getattr(config, "truncated_vocab_size", None) or config.vocab_size)
        self.blocks = nn.ModuleList([_ResidualBlock(H, n_layers) for _ in range(n_heads)])
        self.lm_heads = nn.ModuleList([nn.Linear(H, V, bias=False) for _ in range(n_heads)])

    def forward(self, hidden_states: torch.Tensor) -> list[torch.Tensor]:
        return [b(hidden_states) for b in self.blocks]

    def compute_logits(self, hidden: list[torch.Tensor]) -> list[torch.Tensor]:
        return [head(h) for head, h in zip(self.lm_heads, hidden)]

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        params = dict(self.named_parameters())
        for name, t in weights:
            name = name.replace("medusa_heads.", "")
            # original layout: <head>.<layer>.linear.{weight,bias} and <head>.<n_layers>.weight (the lm head)
            parts = name.split(".")
            if parts[0].isdigit():
                h = parts[0]
                if "linear" in parts:
                    name = f"blocks.{h}.layers.{parts[1]}.{parts[-1]}"
                else:
                    name = f"lm_heads.{h}.{parts[-1]}"
            if name in params:
                params[name].data.copy_(t)
        return None
