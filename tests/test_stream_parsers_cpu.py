"""Streaming tool-call / reasoning parsers: fed in pieces of every size, each machine must produce the same calls
(name, argument string) and the same reasoning / content split as when it reads the finished text at once, must
send a call's name before its arguments end, and must never leak a marker into the content."""
import json

import pytest

from vllm_distributed_amd.entrypoints.openai.parsers import get_reasoning_parser, get_tool_parser
from vllm_distributed_amd.entrypoints.openai.stream_parsers import (ChatDeltaStream, make_reasoning_stream,
                                                                    run_tool_stream, scan_call_objects)

CASES = {
    "hermes": 'Let me check.\n<tool_call>\n{"name": "get_weather", "arguments": {"city": "Paris", "unit": "C"}}\n'
              '</tool_call>\n<tool_call>\n{"name": "f2", "arguments": {}}\n</tool_call>',
    "qwen3": '<tool_call>\n{"name": "get_weather", "arguments": {"city": "Par\\"is"}}\n</tool_call>',
    "llama3_json": '<|python_tag|>{"name": "get_weather", "parameters": {"city": "Paris \\"x\\""}}; '
                   '{"name": "g", "parameters": {"a": [1, 2, {"b": null}]}}',
    "llama4_json": '{"name": "g", "parameters": {"a": 1.5e3}}',
    "mistral": '[TOOL_CALLS][{"name": "add", "arguments": {"a": 1, "b": 2}}, {"name": "mul", "arguments": {"a": 3}}]',
    "pythonic": '[get_weather(city="Paris", n=3), g(x=[1, 2])]',
    "llama4_pythonic": '<|python_start|>[get_weather(city="Pa)ris")]<|python_end|>',
    "granite": '<|tool_call|>[{"name": "a", "arguments": {"x": 1}}]',
    "granite-20b-fc": 'hi <function_call> {"name": "a", "arguments": {"x": 1}}',
    "internlm": 'think<|action_start|><|plugin|>{"name": "a", "parameters": {"x": [1]}}<|action_end|>',
    "jamba": '<tool_calls>[{"name": "a", "arguments": {"x": 1}}]</tool_calls>',
    "minimax": '<tool_calls>\n{"name": "a", "arguments": {"x": 1}}\n{"name": "b", "arguments": {"y": "z"}}\n</tool_calls>',
    "phi4_mini_json": 'functools[{"name": "a", "arguments": {"x": 1}}]',
    "xlam": '```json\n[{"name": "a", "arguments": {"x": 1}}]\n```',
    "deepseek_v3": 'ok<｜tool▁calls▁begin｜><｜tool▁call▁begin｜>function<｜tool▁sep｜>get_weather\n```json\n'
                   '{"city": "Paris"}\n```<｜tool▁call▁end｜><｜tool▁calls▁end｜>',
    "kimi_k2": '<|tool_calls_section_begin|><|tool_call_begin|>functions.get_weather:0<|tool_call_argument_begin|>'
               '{"city": "Paris"}<|tool_call_end|><|tool_calls_section_end|>',
}


def _stream(tool, text, step, reasoning=None):
    st = ChatDeltaStream(reasoning, tool)
    content, reason, calls, order = "", "", {}, []
    for i in range(0, len(text), step):
        d = st.feed(text[i:i + step], final=i + step >= len(text))
        content += d.content
        reason += d.reasoning
        for t in d.tools:
            c = calls.setdefault(t.index, {"name": None, "args": "", "name_at": None})
            if t.name is not None:
                assert c["name"] is None, "a call's name is sent once"
                c["name"], c["name_at"] = t.name, i
            else:
                assert c["name"] is not None, "arguments before the name"
            c["args"] += t.arguments
            order.append(t.index)
    assert order == sorted(order), "calls are streamed in order"
    return content, reason, [(c["name"], c["args"]) for _, c in sorted(calls.items())], calls


@pytest.mark.parametrize("name", sorted(CASES))
def test_streamed_calls_equal_the_whole_text_result(name):
    text = CASES[name]
    whole_content, whole = run_tool_stream(name, text)
    assert whole, name
    # the classic whole-text parser finds the same calls (argument strings equal as JSON)
    _, classic = get_tool_parser(name).extract(text)
    assert [(c.function.name, json.loads(c.function.arguments)) for c in classic] == \
        [(n, json.loads(a)) for n, a in whole]
    for step in (1, 2, 3, 5, 8, 13, 40, 10000):
        content, _, calls, _ = _stream(name, text, step)
        assert calls == whole, (name, step)
        assert content == whole_content, (name, step)
        for marker in ("<tool_call>", "[TOOL_CALLS]", "<|tool_call|>", "<tool_calls>", "<｜tool▁calls▁begin｜>"):
            assert marker not in content


def test_name_goes_out_before_the_arguments_are_complete():
    text = CASES["hermes"]
    _, _, _, calls = _stream("hermes", text, 1)
    first = calls[0]
    assert first["name_at"] < text.index('"unit"')          # sent while the arguments were still being written
    # and argument fragments arrive in more than one piece
    st = ChatDeltaStream(None, "hermes")
    pieces = []
    for i, ch in enumerate(text):
        for t in st.feed(ch, final=i == len(text) - 1).tools:
            if t.index == 0 and t.arguments:
                pieces.append(t.arguments)
    assert len(pieces) > 5 and "".join(pieces) == '{"city": "Paris", "unit": "C"}'


def test_plain_replies_stay_content():
    for name, text in (("hermes", "plain answer with a < sign and <tool nothing"), ("llama3_json", "Hello there"),
                       ("llama3_json", '{"answer": 42}'), ("pythonic", "[1, 2, 3] is a list"),
                       ("mistral", "no calls [TOOL here"), ("deepseek_v3", "just text")):
        for step in (1, 4, 1000):
            content, _, calls, _ = _stream(name, text, step)
            assert calls == [] and content == text, (name, step, content)


@pytest.mark.parametrize("name,text,reason,content", [
    ("deepseek_r1", "<think>step one</think>\nthe answer", "step one", "\nthe answer"),
    ("deepseek_r1", "no opening tag</think>answer", "no opening tag", "answer"),
    ("deepseek_r1", "never closes", "never closes", ""),
    ("qwen3", "<think>\nhm\n</think>\n\nyes", "\nhm\n", "\n\nyes"),
    ("qwen3", "direct answer", "", "direct answer"),
    ("mistral", "[THINK]a[/THINK]b", "a", "b"),
    ("granite", "Here is my thought process: hmm. Here is my response: yes", " hmm. ", " yes"),
    ("granite", "Here's my thought process: x Here's my response: y", " x ", " y"),
    ("hunyuan_a13b", "<think>\nabc\n</think>\n<answer>\nfinal\n</answer>", "\nabc\n", "\n\nfinal\n"),
])
def test_reasoning_split_is_the_same_in_pieces(name, text, reason, content):
    for step in (1, 2, 3, 7, 1000):
        st = make_reasoning_stream(name)
        r = c = ""
        acc = ""
        for i in range(0, len(text), step):
            acc += text[i:i + step]
            dr, dc = st.feed(acc, final=i + step >= len(text))
            r += dr
            c += dc
        assert (r, c) == (reason, content), (name, step, r, c)
    # the whole-text parser agrees up to surrounding white space, where the reply has both markers or none
    cr, cc = get_reasoning_parser(name).extract(text)
    if not (name == "deepseek_r1" and "</think>" not in text):
        assert (cr or "").strip() == reason.strip() and (cc or "").strip() == content.strip()


def test_reasoning_then_tool_call():
    text = '<think>I should call</think>\n<tool_call>{"name": "f", "arguments": {"a": 1}}</tool_call>'
    for step in (1, 3, 9):
        content, reason, calls, _ = _stream("hermes", text, step, reasoning="deepseek_r1")
        assert reason == "I should call" and calls == [("f", '{"a": 1}')] and content.strip() == ""


def test_scanner_on_partial_objects():
    s = '[{"name": "a", "arguments": {"x": "}{"'
    objs = scan_call_objects(s)
    assert len(objs) == 1 and objs[0].name == "a" and not objs[0].done and objs[0].args_end is None
    assert s[objs[0].args_start:] == '{"x": "}{"'
    objs = scan_call_objects('{"arguments": {"k": 1}, "name": "late"} {"name": "b"')
    assert [(o.name, o.done) for o in objs] == [("late", True), ("b", False)]
