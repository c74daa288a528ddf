"""Tool-call and reasoning parsers applied to generated text.

Tool parsers (reference: vllm/entrypoints/openai/tool_parsers/*): hermes
(<tool_call>{json}</tool_call>), llama3_json / llama4_json (bare JSON object with name +
parameters), mistral ([TOOL_CALLS][...]), pythonic / llama4_pythonic ([f(a=1), g()]), granite,
granite-20b-fc, internlm, jamba, phi4_mini_json, xlam, deepseek_v3, kimi_k2, minimax.
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


def _json_array_calls(payload: str) -> list[ToolCall]:
    """Tool calls out of a JSON array (or single object) of {name, arguments | parameters}."""
    try:
        arr = json.loads(payload)
    except json.JSONDecodeError:
        m = re.search(r"\[.*\]|\{.*\}", payload, re.S)
        if not m:
            return []
        try:
            arr = json.loads(m.group(0))
        except json.JSONDecodeError:
            return []
    if isinstance(arr, dict):
        arr = [arr]
    return [_mk(d["name"], d.get("arguments", d.get("parameters", {}))) for d in arr
            if isinstance(d, dict) and "name" in d]


class _TaggedJsonToolParser(ToolParser):
    """Formats that announce a JSON array of calls with a marker: text before the marker stays content."""
    start = ""
    end: Optional[str] = None

    def extract(self, text):
        if self.start not in text:
            return text, []
        head, _, tail = text.partition(self.start)
        if self.end and self.end in tail:
            tail = tail[: tail.index(self.end)]
        calls = _json_array_calls(tail.strip())
        return (head.strip() or None, calls) if calls else (text, [])


class GraniteToolParser(_TaggedJsonToolParser):
    """IBM Granite 3.x: ``<|tool_call|>[{...}]`` (3.1 writes ``<tool_call>[{...}]``)."""
    name = "granite"
    start = "<|tool_call|>"

    def extract(self, text):
        if self.start not in text and text.lstrip().startswith("<tool_call>"):
            text = text.replace("<tool_call>", self.start, 1)
        return super().extract(text)


class Granite20bFCToolParser(ToolParser):
    """granite-20b-functioncalling: one ``<function_call> {json}`` per call."""
    name = "granite-20b-fc"

    def extract(self, text):
        tag = "<function_call>"
        if tag not in text:
            return text, []
        head, *parts = text.split(tag)
        calls = []
        for p in parts:
            calls += _json_array_calls(p.strip())
        return (head.strip() or None, calls) if calls else (text, [])


class InternLMToolParser(ToolParser):
    """InternLM2: ``<|action_start|><|plugin|>{json}<|action_end|>``."""
    name = "internlm"

    def extract(self, text):
        tag = "<|action_start|><|plugin|>"
        if tag not in text:
            return text, []
        head, _, tail = text.partition(tag)
        tail = tail.split("<|action_end|>")[0]
        calls = _json_array_calls(tail.strip())
        return (head.strip() or None, calls) if calls else (text, [])


class JambaToolParser(_TaggedJsonToolParser):
    name = "jamba"
    start, end = "<tool_calls>", "</tool_calls>"


class Phi4MiniJsonToolParser(_TaggedJsonToolParser):
    name = "phi4_mini_json"
    start = "functools"


class XLamToolParser(ToolParser):
    """Salesforce xLAM: a JSON array of calls, bare, fenced in ```json or after ``[TOOL_CALLS]``."""
    name = "xlam"

    def extract(self, text):
        s = text.strip()
        m = re.search(r"```(?:json)?\s*(.*?)```", s, re.S)
        head = ""
        if m:
            head, payload = s[: m.start()], m.group(1)
        elif "[TOOL_CALLS]" in s:
            head, _, payload = s.partition("[TOOL_CALLS]")
        elif s.startswith("[") or s.startswith("{"):
            payload = s
        else:
            return text, []
        calls = _json_array_calls(payload.strip())
        return (head.strip() or None, calls) if calls else (text, [])


class DeepSeekV3ToolParser(ToolParser):
    """DeepSeek-V3 / R1: ``<｜tool▁call▁begin｜>function<｜tool▁sep｜>NAME\n```json\n{args}\n```<｜tool▁call▁end｜>``."""
    name = "deepseek_v3"
    _re = re.compile(r"<｜tool▁call▁begin｜>(?:function)?<｜tool▁sep｜>(?P<name>[^\n]+)\n```json\n(?P<args>.*?)\n```"
                     r"\s*<｜tool▁call▁end｜>", re.S)

    def extract(self, text):
        start = "<｜tool▁calls▁begin｜>"
        if start not in text:
            return text, []
        calls = [_mk(m.group("name").strip(), m.group("args").strip()) for m in self._re.finditer(text)]
        return (text[: text.index(start)].strip() or None, calls) if calls else (text, [])


class KimiK2ToolParser(ToolParser):
    """Kimi-K2: ``<|tool_call_begin|>functions.NAME:IDX<|tool_call_argument_begin|>{args}<|tool_call_end|>``."""
    name = "kimi_k2"
    _re = re.compile(r"<\|tool_call_begin\|>\s*(?:functions\.)?(?P<name>[\w\.\-]+?)(?::\d+)?\s*"
                     r"<\|tool_call_argument_begin\|>\s*(?P<args>.*?)\s*<\|tool_call_end\|>", re.S)

    def extract(self, text):
        start = "<|tool_calls_section_begin|>"
        if start not in text:
            return text, []
        calls = [_mk(m.group("name"), m.group("args")) for m in self._re.finditer(text)]
        return (text[: text.index(start)].strip() or None, calls) if calls else (text, [])


class Llama4PythonicToolParser(PythonicToolParser):
    """Llama 4 wraps the pythonic list in ``<|python_start|> ... <|python_end|>``."""
    name = "llama4_pythonic"

    def extract(self, text):
        s = text.strip()
        s = s.removeprefix("<|python_start|>").removesuffix("<|python_end|>").strip()
        content, calls = super().extract(s)
        return (content, calls) if calls else (text, [])


class Llama4JsonToolParser(Llama3JsonToolParser):
    name = "llama4_json"


class MinimaxToolParser(ToolParser):
    """MiniMax: ``<tool_calls>`` followed by one JSON object per line, closed by ``</tool_calls>``."""
    name = "minimax"

    def extract(self, text):
        m = re.search(r"<tool_calls>(.*?)(?:</tool_calls>|$)", text, re.S)
        if not m:
            return text, []
        calls = []
        for line in m.group(1).strip().splitlines():
            calls += _json_array_calls(line.strip()) if line.strip() else []
        return (text[: m.start()].strip() or None, calls) if calls else (text, [])


TOOL_PARSERS = {c.name: c for c in (HermesToolParser, Llama3JsonToolParser, MistralToolParser, PythonicToolParser,
                                    GraniteToolParser, Granite20bFCToolParser, InternLMToolParser, JambaToolParser,
                                    Phi4MiniJsonToolParser, XLamToolParser, DeepSeekV3ToolParser, KimiK2ToolParser,
                                    Llama4PythonicToolParser, Llama4JsonToolParser, MinimaxToolParser)}
TOOL_PARSERS["qwen3"] = HermesToolParser          # same <tool_call> JSON convention


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


class HunyuanA13BReasoningParser(ReasoningParser):
    """``<think>\n...\n</think>\n<answer>\n...\n</answer>``."""
    name = "hunyuan_a13b"
    _re = re.compile(r"<think>\s*(.*?)\s*</think>\s*(?:<answer>\s*(.*?)\s*(?:</answer>|$))?", re.S)

    def extract(self, text):
        m = self._re.search(text)
        if not m:
            return None, text
        answer = m.group(2) if m.group(2) is not None else text[m.end():]
        return m.group(1).strip() or None, answer.strip() or None


class MistralReasoningParser(ThinkTagReasoningParser):
    """Magistral: ``[THINK] ... [/THINK]``."""
    name = "mistral"
    start, end = "[THINK]", "[/THINK]"


REASONING_PARSERS = {c.name: c for c in (ThinkTagReasoningParser, Qwen3ReasoningParser, GraniteReasoningParser,
                                         HunyuanA13BReasoningParser, MistralReasoningParser)}


def get_reasoning_parser(name: str) -> ReasoningParser:
    if name not in REASONING_PARSERS:
        raise ValueError(f"unknown reasoning parser {name!r}; available: {sorted(REASONING_PARSERS)}")
    return REASONING_PARSERS[name]()
