"""Incremental (streaming) forms of the tool-call and reasoning parsers.

A chat stream must not wait for the end of generation: reasoning text is sent as ``reasoning_content`` deltas
while the model thinks, a tool call's name as soon as it is complete, and its arguments as raw JSON fragments as
they are generated (the concatenation of the fragments is the argument text of the finished call).  Each parser
here is a small state machine fed with the text generated so far; it never re-sends a character and holds back
only what could still turn out to be the beginning of a marker.

The whole-text parsers of ``parsers.py`` are these machines run once over the finished text, so a streamed
response and a non-streamed one carry the same calls and the same argument strings by construction.

Reference: vllm/entrypoints/openai/tool_parsers/abstract_tool_parser.py:61 (``extract_tool_calls_streaming``) and
its per-format implementations (hermes, llama, mistral, pythonic, granite, internlm2, jamba, phi4mini, xlam,
deepseekv3, kimi_k2, minimax, llama4_pythonic), vllm/reasoning/abs_reasoning_parsers.py:100
(``extract_reasoning_content_streaming``), vllm/entrypoints/openai/serving_chat.py:479-1041.
"""
from __future__ import annotations

import ast
import json
from dataclasses import dataclass, field
from typing import Optional

_WS = " \t\r\n"


# ---- a tolerant scanner for JSON that is still being written ----------------------------------------------
def _skip_ws(s: str, i: int) -> int:
    while i < len(s) and s[i] in _WS:
        i += 1
    return i


def _scan_string(s: str, i: int) -> Optional[int]:
    """Index after the closing quote of the string that opens at s[i], or None while it is still open."""
    j = i + 1
    while j < len(s):
        c = s[j]
        if c == "\\":
            j += 2
            continue
        if c == '"':
            return j + 1
        j += 1
    return None


def _scan_value(s: str, i: int) -> Optional[int]:
    """Index after the JSON value that starts at s[i], or None while it is incomplete."""
    if i >= len(s):
        return None
    c = s[i]
    if c == '"':
        return _scan_string(s, i)
    if c in "{[":
        depth, j = 0, i
        while j < len(s):
            c = s[j]
            if c == '"':
                e = _scan_string(s, j)
                if e is None:
                    return None
                j = e
                continue
            if c in "{[":
                depth += 1
            elif c in "}]":
                depth -= 1
                if depth == 0:
                    return j + 1
            j += 1
        return None
    j = i
    while j < len(s) and s[j] not in ",}]" + _WS:
        j += 1
    return j if j < len(s) else None      # a number / literal that touches the end may still grow


@dataclass
class _Obj:
    name: Optional[str] = None
    args_start: Optional[int] = None
    args_end: Optional[int] = None         # exclusive; None while the value is open
    done: bool = False
    end: int = 0


def scan_call_objects(s: str, pos: int = 0, arg_keys=("arguments", "parameters")) -> list[_Obj]:
    """The top-level JSON objects found in s[pos:] (array brackets, commas, tags and other text between them are
    skipped), the last one possibly incomplete.  Of each object: its "name" once that string is closed and the
    extent of its arguments value."""
    out: list[_Obj] = []
    i, n = pos, len(s)
    while i < n:
        if s[i] != "{":
            i += 1
            continue
        o = _Obj()
        j = i + 1
        while True:
            j = _skip_ws(s, j)
            if j >= n:
                break
            if s[j] == "}":
                o.done, j = True, j + 1
                break
            if s[j] == ",":
                j += 1
                continue
            if s[j] != '"':
                j = n
                break
            ke = _scan_string(s, j)
            if ke is None:
                j = n
                break
            try:
                key = json.loads(s[j:ke])
            except json.JSONDecodeError:
                key = None
            j = _skip_ws(s, ke)
            if j >= n:
                break
            if s[j] != ":":
                j = n
                break
            j = _skip_ws(s, j + 1)
            if j >= n:
                break
            ve = _scan_value(s, j)
            if key == "name":
                if ve is not None and s[j] == '"':
                    try:
                        o.name = json.loads(s[j:ve])
                    except json.JSONDecodeError:
                        pass
            elif key in arg_keys and o.args_start is None:
                o.args_start, o.args_end = j, ve
            if ve is None:
                j = n
                break
            j = ve
        o.end = j
        out.append(o)
        if not o.done:
            break
        i = j
    return out


def _held_suffix(text: str, markers) -> int:
    """Length of the longest suffix of ``text`` that is a proper prefix of one of ``markers``: it may be the
    beginning of that marker and must not be sent as content yet."""
    best = 0
    for m in markers:
        for k in range(min(len(m) - 1, len(text)), 0, -1):
            if text.endswith(m[:k]):
                best = max(best, k)
                break
    return best


# ---- events ------------------------------------------------------------------------------------------------
@dataclass
class ToolDelta:
    index: int
    name: Optional[str] = None             # set on the first delta of a call
    arguments: str = ""


@dataclass
class StreamDelta:
    content: str = ""
    reasoning: str = ""
    tools: list[ToolDelta] = field(default_factory=list)

    def __bool__(self) -> bool:
        return bool(self.content or self.reasoning or self.tools)


# ---- tool calls --------------------------------------------------------------------------------------------
class ToolCallStream:
    """Base: ``feed(text_so_far, final)`` returns the new content and the new tool-call deltas."""

    def __init__(self):
        self.sent_content = 0
        self.sent: list[dict] = []          # per call: {"name": sent?, "args": characters sent}

    def _emit_call(self, out: list[ToolDelta], k: int, name: Optional[str], args: str, closed: bool) -> None:
        """Send what is new of call k: the name once it is known, then the unsent tail of ``args``."""
        while len(self.sent) <= k:
            self.sent.append({"name": False, "args": 0})
        st = self.sent[k]
        if name is None:
            return                          # arguments wait for the name: a client needs it first
        d = ToolDelta(index=k)
        if not st["name"]:
            st["name"], d.name = True, name
        if closed and not args and st["args"] == 0:
            args = "{}"
        if len(args) > st["args"]:
            d.arguments = args[st["args"]:]
            st["args"] = len(args)
        if d.name is not None or d.arguments:
            out.append(d)

    def _content(self, region: str, final: bool, markers=()) -> str:
        hold = 0 if final else _held_suffix(region, markers)
        upto = len(region) - hold
        new = region[self.sent_content:upto] if upto > self.sent_content else ""
        self.sent_content = max(self.sent_content, upto)
        return new

    def feed(self, text: str, final: bool = False) -> tuple[str, list[ToolDelta]]:
        raise NotImplementedError


class JsonCallStream(ToolCallStream):
    """Formats whose calls are JSON objects {name, arguments | parameters}: after a marker (hermes, mistral, granite,
    jamba, internlm, phi4-mini, minimax, granite-20b-fc, xlam) and / or as the whole reply (llama3 / llama4 JSON,
    xlam).  Text in front of the first marker is content."""

    def __init__(self, markers=(), bare: bool = False, prefixes=(), require_args_key: bool = False):
        super().__init__()
        self.markers, self.bare, self.prefixes, self.require_args_key = tuple(markers), bare, tuple(prefixes), require_args_key
        self.payload_at: Optional[int] = None
        self.content_end: Optional[int] = None
        self.is_content = False

    def _locate(self, text: str, final: bool) -> bool:
        """Find where the calls begin.  Returns False while that cannot be decided yet (a bare reply of which only
        white space or a partial prefix has arrived)."""
        if self.payload_at is not None or self.is_content:
            return True
        best = None
        for m in self.markers:
            p = text.find(m)
            if p >= 0 and (best is None or p < best[0]):
                best = (p, len(m))
        if best is not None:
            self.content_end, self.payload_at = best[0], best[0] + best[1]
            return True
        if self.bare:
            i = _skip_ws(text, 0)
            for pre in self.prefixes:                   # e.g. <|python_tag|>
                if text.startswith(pre, i):
                    i = _skip_ws(text, i + len(pre))
                elif not final and i < len(text) and pre.startswith(text[i:]):
                    return False                        # may still become the prefix
            if i >= len(text):
                return final
            if text[i] in "{[":
                self.content_end, self.payload_at = 0, i
            elif not self.markers:
                self.is_content = True
        return True

    def feed(self, text, final=False):
        tools: list[ToolDelta] = []
        if not self._locate(text, final):
            return "", tools
        if self.payload_at is None:
            return self._content(text, final, self.markers), tools
        content = self._content(text[: self.content_end], True)
        k = 0
        for o in scan_call_objects(text, self.payload_at):
            if o.done and (o.name is None or (self.require_args_key and o.args_start is None)):
                continue                                # a JSON object that is not a call
            if self.require_args_key and o.args_start is None:
                break                                   # not yet known to be a call
            raw = "" if o.args_start is None else text[o.args_start: o.args_end if o.args_end is not None else len(text)]
            self._emit_call(tools, k, o.name, raw, o.done)
            k += 1
        if final and self.content_end == 0 and not any(st["name"] for st in self.sent):
            # a bare JSON reply that held no call after all: it is the answer itself
            content = self._content(text, True)
        return content, tools


class DelimitedCallStream(ToolCallStream):
    """Formats that spell a call as BEGIN name NAME_END arguments ARGS_END (DeepSeek-V3, Kimi-K2)."""

    def __init__(self, section: str, begin: str, name_end: str, args_end: str, name_clean=lambda s: s.strip()):
        super().__init__()
        self.section, self.begin, self.name_end, self.args_end, self.name_clean = section, begin, name_end, args_end, name_clean

    def feed(self, text, final=False):
        tools: list[ToolDelta] = []
        p = text.find(self.section)
        if p < 0:
            return self._content(text, final, (self.section,)), tools
        content = self._content(text[:p], True)
        i, k = p + len(self.section), 0
        while True:
            b = text.find(self.begin, i)
            if b < 0:
                break
            ns = b + len(self.begin)
            ne = text.find(self.name_end, ns)
            if ne < 0:
                break
            name = self.name_clean(text[ns:ne])
            a0 = ne + len(self.name_end)
            ae = text.find(self.args_end, a0)
            if ae < 0:
                hold = 0 if final else _held_suffix(text[a0:], (self.args_end,))
                self._emit_call(tools, k, name, text[a0: len(text) - hold].lstrip(), False)
                break
            self._emit_call(tools, k, name, text[a0:ae].strip(), True)
            k += 1
            i = ae + len(self.args_end)
        return content, tools


class PythonicCallStream(ToolCallStream):
    """``[f(a=1, b="x"), g()]``: a call is sent when its closing parenthesis has arrived (name, then the arguments
    as one JSON fragment)."""

    def __init__(self, prefixes=(), suffixes=()):
        super().__init__()
        self.prefixes, self.suffixes = tuple(prefixes), tuple(suffixes)
        self.decided: Optional[bool] = None

    def feed(self, text, final=False):
        tools: list[ToolDelta] = []
        i = _skip_ws(text, 0)
        for pre in self.prefixes:
            if text.startswith(pre, i):
                i = _skip_ws(text, i + len(pre))
            elif not final and i < len(text) and pre.startswith(text[i:]):
                return "", tools
        if self.decided is None:
            if i >= len(text):
                if not final:
                    return "", tools
                self.decided = False
            else:
                self.decided = text[i] == "["
        if not self.decided:
            return self._content(text, final), tools
        # top-level calls inside the brackets
        j, k = i + 1, 0
        n = len(text)
        while j < n:
            j = _skip_ws(text, j)
            if j >= n or text[j] == "]":
                break
            if text[j] == ",":
                j += 1
                continue
            start, depth, q = j, 0, None
            end = None
            if not (text[j].isalpha() or text[j] == "_"):
                if not any(st["name"] for st in self.sent):
                    self.decided = False                 # a list, but not of calls: plain content
                    return self._content(text, final), tools
                break
            while j < n:
                c = text[j]
                if depth == 0 and not q and c in ",]":
                    break                                # an element without parentheses: not a call
                if q:
                    if c == "\\":
                        j += 1
                    elif c == q:
                        q = None
                elif c in "\"'":
                    q = c
                elif c in "([{":
                    depth += 1
                elif c in ")]}":
                    depth -= 1
                    if depth == 0 and c == ")":
                        end = j + 1
                        break
                    if depth < 0:
                        break
                j += 1
            if end is None:
                if (j < n or final) and not any(st["name"] for st in self.sent):
                    self.decided = False
                    return self._content(text, final), tools
                break
            try:
                node = ast.parse(text[start:end], mode="eval").body
                if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Name)):
                    raise ValueError
                args = {kw.arg: ast.literal_eval(kw.value) for kw in node.keywords}
                self._emit_call(tools, k, node.func.id, json.dumps(args, ensure_ascii=False), True)
                k += 1
            except (SyntaxError, ValueError):
                if k == 0 and not self.sent:
                    self.decided = False                 # not a call list after all: plain content
                    return self._content(text, final), tools
                break
            j = end
        return "", tools


def make_tool_stream(name: str) -> ToolCallStream:
    """The streaming machine of tool parser ``name`` (same names as parsers.TOOL_PARSERS)."""
    if name in ("hermes", "qwen3"):
        return JsonCallStream(markers=("<tool_call>",))
    if name in ("llama3_json", "llama4_json"):
        return JsonCallStream(bare=True, prefixes=("<|python_tag|>",), require_args_key=True)
    if name == "mistral":
        return JsonCallStream(markers=("[TOOL_CALLS]",))
    if name == "granite":
        return JsonCallStream(markers=("<|tool_call|>", "<tool_call>"))
    if name == "granite-20b-fc":
        return JsonCallStream(markers=("<function_call>",))
    if name == "internlm":
        return JsonCallStream(markers=("<|action_start|><|plugin|>",))
    if name == "jamba":
        return JsonCallStream(markers=("<tool_calls>",))
    if name == "minimax":
        return JsonCallStream(markers=("<tool_calls>",))
    if name == "phi4_mini_json":
        return JsonCallStream(markers=("functools",))
    if name == "xlam":
        return JsonCallStream(markers=("```json", "```", "[TOOL_CALLS]"), bare=True)
    if name == "deepseek_v3":
        return DelimitedCallStream("<｜tool▁calls▁begin｜>", "<｜tool▁call▁begin｜>", "\n```json\n", "\n```",
                                   name_clean=lambda s: s.split("<｜tool▁sep｜>")[-1].strip())
    if name == "kimi_k2":
        def clean(s: str) -> str:
            s = s.strip()
            s = s[len("functions."):] if s.startswith("functions.") else s
            head, _, idx = s.rpartition(":")
            return head if head and idx.isdigit() else s
        return DelimitedCallStream("<|tool_calls_section_begin|>", "<|tool_call_begin|>", "<|tool_call_argument_begin|>",
                                   "<|tool_call_end|>", name_clean=clean)
    if name == "pythonic":
        return PythonicCallStream()
    if name == "llama4_pythonic":
        return PythonicCallStream(prefixes=("<|python_start|>",), suffixes=("<|python_end|>",))
    raise ValueError(f"no streaming form for tool parser {name!r}")


def run_tool_stream(name: str, text: str) -> tuple[str, list[tuple[str, str]]]:
    """The machine run once over a finished text: (content, [(name, arguments)])."""
    st = make_tool_stream(name)
    content, deltas = st.feed(text, final=True)
    calls: dict[int, list] = {}
    for d in deltas:
        c = calls.setdefault(d.index, ["", ""])
        if d.name is not None:
            c[0] = d.name
        c[1] += d.arguments
    return content, [(c[0], c[1]) for _, c in sorted(calls.items())]


# ---- reasoning ---------------------------------------------------------------------------------------------
class ReasoningStream:
    """Splits the text into a reasoning part and a content part around start / end markers.

    ``implicit``: the reply begins inside the reasoning even without a start marker (DeepSeek-R1 chat templates
    put ``<think>`` into the prompt), so everything up to the end marker is reasoning.  Otherwise the reply is
    reasoning only if it begins (after white space) with a start marker.  ``strip``: markers removed from the
    content part (``<answer>`` wrappers)."""

    def __init__(self, starts, ends, implicit: bool, strip=()):
        self.starts, self.ends, self.implicit, self.strip = tuple(starts), tuple(ends), implicit, tuple(strip)
        self.state = "init"
        self.body_at = 0                    # where the reasoning text begins
        self.content_at = 0
        self.sent_r = 0
        self.sent_c = 0

    def feed(self, text: str, final: bool = False) -> tuple[str, str]:
        """(new reasoning text, new content text) given the whole text generated so far."""
        if self.state == "init":
            i = _skip_ws(text, 0)
            hit = next((m for m in self.starts if text.startswith(m, i)), None)
            if hit is not None:
                self.state, self.body_at = "reasoning", i + len(hit)
            elif not final and i < len(text) and any(m.startswith(text[i:]) for m in self.starts):
                return "", ""               # may still become a start marker
            elif i >= len(text) and not final:
                return "", ""
            else:
                self.state = "reasoning" if self.implicit else "content"
                self.body_at = self.content_at = 0
        r_new = c_new = ""
        if self.state == "reasoning":
            body = text[self.body_at:]
            cut = None
            for m in self.ends:
                p = body.find(m)
                if p >= 0 and (cut is None or p < cut[0]):
                    cut = (p, len(m))
            if cut is None:
                hold = 0 if final else _held_suffix(body, self.ends)
                upto = len(body) - hold
            else:
                upto = cut[0]
            if upto > self.sent_r:
                r_new = body[self.sent_r:upto]
                self.sent_r = upto
            if cut is not None:
                self.state = "content"
                self.content_at = self.body_at + cut[0] + cut[1]
        if self.state == "content":
            body = text[self.content_at:]
            for m in self.strip:
                body = body.replace(m, "")
            hold = 0 if final else _held_suffix(body, self.strip)
            upto = len(body) - hold
            if upto > self.sent_c:
                c_new = body[self.sent_c:upto]
                self.sent_c = upto
        return r_new, c_new


def make_reasoning_stream(name: str) -> ReasoningStream:
    if name == "deepseek_r1":
        return ReasoningStream(("<think>",), ("</think>",), implicit=True)
    if name == "qwen3":
        return ReasoningStream(("<think>",), ("</think>",), implicit=False)
    if name == "mistral":
        return ReasoningStream(("[THINK]",), ("[/THINK]",), implicit=False)
    if name == "granite":
        return ReasoningStream(("Here is my thought process:", "Here's my thought process:"),
                               ("Here is my response:", "Here's my response:"), implicit=False)
    if name == "hunyuan_a13b":
        return ReasoningStream(("<think>",), ("</think>",), implicit=False, strip=("<answer>", "</answer>"))
    raise ValueError(f"no streaming form for reasoning parser {name!r}")


# ---- one chat choice ------------------------------------------------------------------------------------------
class ChatDeltaStream:
    """Reasoning and tool-call machines chained for one streamed choice: feed it the newly generated text, get
    the deltas to send."""

    def __init__(self, reasoning: Optional[str] = None, tool: Optional[str] = None):
        self.reasoning = make_reasoning_stream(reasoning) if reasoning else None
        self.tool = make_tool_stream(tool) if tool else None
        self.text = ""
        self.content = ""                   # content part (after the reasoning split), fed to the tool machine
        self.num_tool_calls = 0

    def feed(self, new_text: str, final: bool = False) -> StreamDelta:
        self.text += new_text
        out = StreamDelta()
        if self.reasoning is not None:
            out.reasoning, c_new = self.reasoning.feed(self.text, final)
        else:
            c_new = new_text
        self.content += c_new
        if self.tool is not None:
            out.content, out.tools = self.tool.feed(self.content, final)
            self.num_tool_calls = sum(1 for s in self.tool.sent if s["name"])
        else:
            out.content = c_new
        return out
