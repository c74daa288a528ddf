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
from .llama import LlamaDecoderLayer
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
