"""More draft networks for speculative decoding: the multi-token-prediction layers that DeepSeek-V3 and MiMo ship
inside their own checkpoints, and IBM's MLP speculators.

The MTP drafts follow the EAGLE calling convention (``model(input_ids, positions, target_hidden) -> hidden``,
``compute_logits``), so EagleProposer drives them: one decoder layer over a projection of [token embedding ;
target hidden state], its KV cached in the request's own blocks.  One MTP layer is reused for every drafted position.
Reference: vllm/model_executor/models/{deepseek_mtp,mimo_mtp,mlp_speculator}.py.
"""
from __future__ import annotations

import math
import re
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from ..layers.basic import LogitsProcessor, ParallelLMHead, RMSNorm, VocabParallelEmbedding
from ..layers.linear import ReplicatedLinear
from ..layers.moe import FusedMoE
from .utils import load_weights_with_mapping


class _MTPDraft(nn.Module):
    share_embeddings = True
    hidden_first = False          # order of the two halves of the projection's input

    def _common(self, config):
        H = config.hidden_size
        eps = getattr(config, "rms_norm_eps", 1e-6)
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, H, prefix="draft.embed_tokens")
        self.token_norm = RMSNorm(H, eps)
        self.hidden_norm = RMSNorm(H, eps)
        self.in_proj = ReplicatedLinear(2 * H, H, bias=False, prefix="draft.in_proj")
        self.out_norm = RMSNorm(H, eps)
        self.lm_head = ParallelLMHead(config.vocab_size, H, prefix="draft.lm_head")
        self.logits_processor = LogitsProcessor(config.vocab_size, config.vocab_size)
        self._own_embed = self._own_head = False

    def set_shared(self, embed, lm_head, logits_processor) -> None:
        if not self._own_embed:
            self.embed_tokens = embed
        if not self._own_head:
            self.lm_head, self.logits_processor = lm_head, logits_processor

    def forward(self, input_ids, positions, hidden_states):
        e = self.embed_tokens(input_ids)
        e = e * (positions != 0).unsqueeze(-1).to(e.dtype)           # position 0 has no predecessor to draft from
        e, h = self.token_norm(e), self.hidden_norm(hidden_states)
        x = self.in_proj(torch.cat((h, e) if self.hidden_first else (e, h), dim=-1))
        x, residual = self.block(positions, x, None)
        return self._finish(x + residual)

    def _finish(self, x):
        return x

    def compute_logits(self, hidden_states):
        return self.logits_processor(self.lm_head, self.out_norm(hidden_states)) if self.norm_in_head \
            else self.logits_processor(self.lm_head, hidden_states)


class DeepSeekMTPModel(_MTPDraft):
    """``model.layers.<num_hidden_layers>`` of a DeepSeek-V3 checkpoint: enorm / hnorm / eh_proj, a full decoder
    layer (latent attention + experts) and a head with its own norm."""
    norm_in_head = True

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        from .deepseek import DeepseekDecoderLayer
        config = vllm_config.model_config.hf_config
        self.config = config
        self._common(config)
        self.layer_index = int(config.num_hidden_layers)
        self.block = DeepseekDecoderLayer(config, vllm_config, getattr(vllm_config, "quant_config", None),
                                          f"draft.layers.{self.layer_index}")

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        pre = f"model.layers.{self.layer_index}."
        table = (("enorm.", "token_norm."), ("hnorm.", "hidden_norm."), ("eh_proj.", "in_proj."),
                 ("shared_head.norm.", "out_norm."), ("shared_head.head.", "lm_head."), ("embed_tokens.", "embed_tokens."))
        experts = FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj",
                                                      int(getattr(self.config, "n_routed_experts", 0) or 0))

        def rename(name: str):
            if not name.startswith(pre):
                return None
            name = name[len(pre):]
            for a, b in table:
                if name.startswith(a):
                    if a == "embed_tokens.":
                        self._own_embed = True
                    if a == "shared_head.head.":
                        self._own_head = True
                    return b + name[len(a):]
            return "block." + name
        stacked = [(".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]
        load_weights_with_mapping(self, weights, stacked, rename=rename, expert_mapping=experts)
        return None


class MiMoMTPModel(_MTPDraft):
    """``model.mtp_layers.0`` of a MiMo checkpoint: token / hidden norms, input_proj over [hidden ; embedding], one
    Qwen2 layer and a final norm; the head is the target's."""
    norm_in_head = False
    hidden_first = True

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        from .llama import LlamaDecoderLayer

        class Qwen2Block(LlamaDecoderLayer):                          # biases on q, k, v only
            def _attention_extras(self, config, layer_idx: int) -> dict:
                return {"bias": True, "o_bias": False}
        config = vllm_config.model_config.hf_config
        self.config = config
        self._common(config)
        self.block = Qwen2Block(config, vllm_config, None, f"draft.layers.{int(config.num_hidden_layers)}")

    def _finish(self, x):
        return self.out_norm(x)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        table = (("token_layernorm.", "token_norm."), ("hidden_layernorm.", "hidden_norm."), ("input_proj.", "in_proj."),
                 ("final_layernorm.", "out_norm."))

        def rename(name: str):
            m = re.match(r"model\.mtp_layers\.(\d+)\.(.*)", name)
            if not m or m.group(1) != "0":
                return None
            name = m.group(2)
            for a, b in table:
                if name.startswith(a):
                    return b + name[len(a):]
            return "block." + name
        stacked = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
                   (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]
        load_weights_with_mapping(self, weights, stacked, rename=rename)
        return None


# ---- MLP speculator ------------------------------------------------------------------------------------------
class _ScaleNorm(nn.Module):
    def __init__(self, n: int, affine: bool = True, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(n), requires_grad=False) if affine else None
        self.bias = nn.Parameter(torch.zeros(n), requires_grad=False) if affine else None

    def forward(self, x):
        x = x * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + self.eps).to(x.dtype)
        return x if self.weight is None else x * self.weight + self.bias


class MLPSpeculatorPreTrainedModel(nn.Module):
    """Stage i turns (state, last token) into the next state and a guess for token t + 1 + i:
    ``state = gelu(ln_i(proj_i(state) + w * emb_i(token)))``, ``logits = head_i(state)``."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        c = vllm_config.model_config.hf_config
        self.config = c
        self.n_predict = int(c.n_predict)
        emb_dim = int(c.emb_dim)
        inner = int(getattr(c, "inner_dim", 0) or emb_dim)
        n = int(getattr(c, "num_lookahead_tokens", None) or self.n_predict)
        tie = bool(getattr(c, "tie_weights", False))
        self.scale_input = bool(getattr(c, "scale_input", False))

        def stages(make, first=None):
            if tie:
                shared = make(1)
                return nn.ModuleList(([first] if first is not None else [shared]) + [shared] * (n - 1))
            return nn.ModuleList([first if (i == 0 and first is not None) else make(i) for i in range(n)])
        self.emb = stages(lambda i: nn.Embedding(c.vocab_size, inner))
        self.proj = stages(lambda i: nn.Linear(inner, inner, bias=False), first=nn.Linear(emb_dim, inner, bias=False))
        self.head = stages(lambda i: nn.Linear(inner, c.vocab_size, bias=False))
        self.ln = stages(lambda i: _ScaleNorm(inner))
        if self.scale_input:
            self.ln0 = _ScaleNorm(emb_dim, affine=False)
        self.state_weight = 0.5 ** (0.5 / self.n_predict)
        self.emb_weight = math.sqrt((1 - self.state_weight ** 2) * (inner / 2))

    @torch.inference_mode()
    def propose(self, last_tokens: torch.Tensor, hidden: torch.Tensor, k: int) -> torch.Tensor:
        """last_tokens [S], hidden [S, emb_dim] -> drafted ids [S, k] (greedy)."""
        if k > len(self.emb):
            raise ValueError(f"this speculator drafts at most {len(self.emb)} tokens, {k} requested")
        state = hidden
        if self.scale_input:
            state = self.ln0(state) / math.sqrt(2)
        tok, out = last_tokens, []
        for i in range(k):
            z = self.emb[i](tok)
            state = self.proj[i](state) + z * (self.emb_weight / self.state_weight)
            state = F.gelu(self.ln[i](state))
            tok = self.head[i](state).argmax(dim=-1)
            out.append(tok)
        return torch.stack(out, dim=1)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        params = dict(self.named_parameters())
        for name, t in weights:
            name = name.replace("speculator.", "")
            if name in params:
                params[name].data.copy_(t)
        return None
