"""Tool-call and reasoning parsers applied to generated text.

Tool parsers (reference: vllm/entrypoints/openai/tool_parsers/*): hermes
(<tool_call>{json}</tool_call>), llama3_json (bare JSON object with name +
parameters), mistral ([TOOL_CALLS][...]), pythonic ([f(a=1), g()]).
Reasoning parsers (reference: vllm/reasoning/*): deepseek_r1 / qwen3
(<think>...</think>), granite ("Here is my thought process:").
"""
from __future__ import annotations

import ast
import json
import re
from typing import Optional

from .protocol import FunctionCall, ToolCall


class ToolParser:
    name = "base"

    def extract(self, text: str) -> tuple[Optional[str], list[ToolCall]]:
        """(remaining content, tool calls)."""
        raise NotImplementedError


def _mk(name: str, args) -> ToolCall:
    return ToolCall(function=FunctionCall(name=name, arguments=args if isinstance(args, str)
                                          else json.dumps(args, ensure_ascii=False)))


class HermesToolParser(ToolParser):
    name = "hermes"
    _re = re.compile(r"<tool_call>\s*(.*?)\s*(?:</tool_call>|$)", re.S)

    def extract(self, text):
        calls = []
        for m in self._re.finditer(text):
            try:
                d = json.loads(m.group(1))
                calls.append(_mk(d["name"], d.get("arguments", d.get("parameters", {}))))
            except (json.JSONDecodeError, KeyError, TypeError):
                continue
        if not calls:
            return text, []
        content = text[: text.find("<tool_call>")].strip()
        return content or None, calls


class Llama3JsonToolParser(ToolParser):
    name = "llama3_json"

    def extract(self, text):
        s = text.strip()
        if s.startswith("<|python_tag|>"):
            s = s[len("<|python_tag|>"):]
        calls = []
        dec = json.JSONDecoder()
        i = 0
        while i < len(s):
            if s[i] != "{":
                i += 1
                continue
            try:
                obj, end = dec.raw_decode(s, i)
            except json.JSONDecodeError:
                break
            if isinstance(obj, dict) and "name" in obj and ("parameters" in obj or "arguments" in obj):
                calls.append(_mk(obj["name"], obj.get("parameters", obj.get("arguments"))))
            i = end
        return (None, calls) if calls else (text, [])


class MistralToolParser(ToolParser):
    name = "mistral"

    def extract(self, text):
        tag = "[TOOL_CALLS]"
        if tag not in text:
            return text, []
        head, _, tail = text.partition(tag)
        try:
            arr = json.loads(tail.strip())
        except json.JSONDecodeError:
            m = re.search(r"\[.*\]", tail, re.S)
            if not m:
                return text, []
            try:
                arr = json.loads(m.group(0))
            except json.JSONDecodeError:
                return text, []
        calls = [_mk(d["name"], d.get("arguments", {})) for d in arr if isinstance(d, dict) and "name" in d]
        return head.strip() or None, calls


class PythonicToolParser(ToolParser):
    name = "pythonic"

    def extract(self, text):
        s = text.strip()
        if not (s.startswith("[") and s.endswith("]")):
            return text, []
        try:
            tree = ast.parse(s, mode="eval")
        except SyntaxError:
            return text, []
        if not isinstance(tree.body, ast.List):
            return text, []
        calls = []
        for el in tree.body.elts:
            if not (isinstance(el, ast.Call) and isinstance(el.func, ast.Name)):
                return text, []
            try:
                args = {kw.arg: ast.literal_eval(kw.value) for kw in el.keywords}
            except (ValueError, SyntaxError):
                return text, []
            calls.append(_mk(el.func.id, args))
        return None, calls


TOOL_PARSERS = {c.name: c for c in (HermesToolParser, Llama3JsonToolParser, MistralToolParser, PythonicToolParser)}


def get_tool_parser(name: str) -> ToolParser:
    if name not in TOOL_PARSERS:
        raise ValueError(f"unknown tool parser {name!r}; available: {sorted(TOOL_PARSERS)}")
    return TOOL_PARSERS[name]()


class ReasoningParser:
    name = "base"

    def extract(self, text: str) -> tuple[Optional[str], Optional[str]]:
        """(reasoning_content, content)."""
        raise NotImplementedError


class ThinkTagReasoningParser(ReasoningParser):
    name = "deepseek_r1"
    start, end = "<think>", "</think>"

    def extract(self, text):
        if self.end not in text:
            # DeepSeek-R1 may omit the opening tag; without the closing tag everything is reasoning
            return (text.replace(self.start, "").strip() or None, None) if self.start in text else (None, text)
        head, _, tail = text.partition(self.end)
        head = head.replace(self.start, "")
        return head.strip() or None, tail.strip() or None


class Qwen3ReasoningParser(ThinkTagReasoningParser):
    name = "qwen3"

    def extract(self, text):
        if self.start not in text or self.end not in text:
            return None, text
        return super().extract(text)


class GraniteReasoningParser(ReasoningParser):
    name = "granite"
    _re = re.compile(r"(?:Here is my thought process:|Here's my thought process:)(.*?)"
                     r"(?:Here is my response:|Here's my response:)(.*)", re.S)

    def extract(self, text):
        m = self._re.search(text)
        if not m:
            return None, text
        return m.group(1).strip() or None, m.group(2).strip() or None


REASONING_PARSERS = {c.name: c for c in (ThinkTagReasoningParser, Qwen3ReasoningParser, GraniteReasoningParser)}


def get_reasoning_parser(name: str) -> ReasoningParser:
    if name not in REASONING_PARSERS:
        raise ValueError(f"unknown reasoning parser {name!r}; available: {sorted(REASONING_PARSERS)}")
    return REASONING_PARSERS[name]()
