"""Prompt types accepted by the public API (reference: vllm/inputs/data.py)."""
from __future__ import annotations

from typing import Any, Optional, TypedDict, Union


class TextPrompt(TypedDict, total=False):
    prompt: str
    multi_modal_data: Optional[dict]
    mm_processor_kwargs: Optional[dict]
    cache_salt: Optional[str]


class TokensPrompt(TypedDict, total=False):
    prompt_token_ids: list[int]
    prompt: Optional[str]
    multi_modal_data: Optional[dict]
    cache_salt: Optional[str]


class EmbedsPrompt(TypedDict, total=False):
    prompt_embeds: Any
    cache_salt: Optional[str]


PromptType = Union[str, TextPrompt, TokensPrompt, EmbedsPrompt]


def parse_prompt(p: PromptType) -> dict:
    if isinstance(p, str):
        return {"prompt": p}
    if isinstance(p, dict):
        if "prompt" not in p and "prompt_token_ids" not in p and "prompt_embeds" not in p:
            raise ValueError("a prompt dict needs 'prompt', 'prompt_token_ids' or 'prompt_embeds'")
        return dict(p)
    if isinstance(p, (list, tuple)) and all(isinstance(x, int) for x in p):
        return {"prompt_token_ids": list(p)}
    raise TypeError(f"unsupported prompt type {type(p).__name__}")
