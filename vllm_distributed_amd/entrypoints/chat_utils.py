"""Chat templating and content-part handling (reference: vllm/entrypoints/chat_utils.py)."""
from __future__ import annotations

import json
from typing import Any, Optional

_FALLBACK_TEMPLATE = (
    "{% for m in messages %}<|{{ m['role'] }}|>\n{{ m['content'] }}\n{% endfor %}"
    "{% if add_generation_prompt %}<|assistant|>\n{% endif %}")


IMAGE_PLACEHOLDER = "<image>"       # replaced by the model's image token string before tokenisation
AUDIO_PLACEHOLDER = "<audio>"


def load_image(url: str):
    """``data:`` URLs (base64) and local ``file://`` paths; there is no network fetch on purpose."""
    import base64
    import io

    from PIL import Image
    if url.startswith("data:"):
        head, _, payload = url.partition(",")
        raw = base64.b64decode(payload) if ";base64" in head else payload.encode()
        return Image.open(io.BytesIO(raw)).convert("RGB")
    if url.startswith("file://"):
        return Image.open(url[len("file://"):]).convert("RGB")
    raise ValueError("image_url must be a data: URL or a file:// path (remote fetching is disabled)")


def _decode_wav(raw: bytes):
    """WAV bytes -> (mono float32 waveform in [-1, 1], sampling rate)."""
    import io

    import numpy as np
    from scipy.io import wavfile
    rate, data = wavfile.read(io.BytesIO(raw))
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":                    # 8-bit PCM is unsigned
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    return data, int(rate)


def load_audio(url: str):
    """``data:`` URLs (base64 WAV) and local ``file://`` paths to WAV files."""
    import base64
    if url.startswith("data:"):
        head, _, payload = url.partition(",")
        return _decode_wav(base64.b64decode(payload) if ";base64" in head else payload.encode())
    if url.startswith("file://"):
        with open(url[len("file://"):], "rb") as f:
            return _decode_wav(f.read())
    raise ValueError("audio_url must be a data: URL or a file:// path (remote fetching is disabled)")


def load_media(parts: list[dict]) -> dict:
    """Media content parts -> ``multi_modal_data``."""
    import base64
    images, clips = [], []
    for part in parts:
        t = part.get("type")
        if t == "image_url":
            u = part["image_url"]
            images.append(load_image(u["url"] if isinstance(u, dict) else u))
        elif t == "audio_url":
            u = part["audio_url"]
            clips.append(load_audio(u["url"] if isinstance(u, dict) else u))
        elif t == "input_audio":                        # OpenAI form: {"data": base64, "format": "wav"}
            a = part["input_audio"]
            if str(a.get("format", "wav")).lower() != "wav":
                raise ValueError("input_audio: only WAV data can be decoded here")
            clips.append(_decode_wav(base64.b64decode(a["data"])))
        else:
            raise ValueError(f"{t} inputs are not supported by this model")
    out = {}
    if images:
        out["image"] = images
    if clips:
        out["audio"] = clips
    return out


def _flatten_content(content: Any) -> tuple[str, list[dict]]:
    """OpenAI content may be a string or a list of typed parts; returns (text, media parts).  Each image part
    leaves ``IMAGE_PLACEHOLDER`` in the text at the position where it occurred."""
    if content is None:
        return "", []
    if isinstance(content, str):
        return content, []
    texts, media = [], []
    for part in content:
        if isinstance(part, str):
            texts.append(part)
            continue
        t = part.get("type")
        if t in ("text", "input_text"):
            texts.append(part.get("text", ""))
        elif t in ("image_url", "audio_url", "video_url", "input_audio", "image_embeds"):
            media.append(part)
            if t == "image_url":
                texts.append(IMAGE_PLACEHOLDER)
            elif t in ("audio_url", "input_audio"):
                texts.append(AUDIO_PLACEHOLDER)
        elif t == "refusal":
            texts.append(part.get("refusal", ""))
        else:
            raise ValueError(f"unknown content part type {t!r}")
    return "\n".join(texts), media


def parse_chat_messages(messages: list[dict]) -> tuple[list[dict], list[dict]]:
    conv, media_all = [], []
    for m in messages:
        if "role" not in m:
            raise ValueError("every message needs a role")
        text, media = _flatten_content(m.get("content"))
        out = {"role": m["role"], "content": text}
        if m.get("tool_calls"):
            tcs = []
            for tc in m["tool_calls"]:
                fn = dict(tc.get("function", {}))
                if isinstance(fn.get("arguments"), str):
                    try:
                        fn["arguments"] = json.loads(fn["arguments"])
                    except json.JSONDecodeError:
                        pass
                tcs.append({**tc, "function": fn})
            out["tool_calls"] = tcs
        for k in ("name", "tool_call_id"):
            if k in m:
                out[k] = m[k]
        conv.append(out)
        media_all += media
    return conv, media_all


def apply_chat_template(tokenizer, messages: list[dict], chat_template: Optional[str] = None,
                        add_generation_prompt: bool = True, continue_final_message: bool = False,
                        tools: Optional[list[dict]] = None, **kwargs) -> str:
    conv, _ = parse_chat_messages(messages)
    template = chat_template or getattr(tokenizer, "chat_template", None)
    if template is None:
        from jinja2 import Template
        return Template(_FALLBACK_TEMPLATE).render(messages=conv, add_generation_prompt=add_generation_prompt)
    return tokenizer.apply_chat_template(conv, chat_template=chat_template, tokenize=False,
                                         add_generation_prompt=add_generation_prompt,
                                         continue_final_message=continue_final_message, tools=tools, **kwargs)
