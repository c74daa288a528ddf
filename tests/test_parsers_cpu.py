"""Tool-call and reasoning parsers on canned model outputs (the formats of the reference's
tests/entrypoints/openai/tool_parsers/* and tests/reasoning/*)."""
import json

import pytest

from vllm_distributed_amd.entrypoints.openai.parsers import (REASONING_PARSERS, TOOL_PARSERS, get_reasoning_parser,
                                                             get_tool_parser)

CALL = {"name": "get_weather", "arguments": {"city": "Paris", "days": 2}}
J = json.dumps(CALL)

CASES = {
    "hermes": f"Let me check.<tool_call>\n{J}\n</tool_call>",
    "qwen3": f"Let me check.<tool_call>\n{J}\n</tool_call>",
    "llama3_json": '<|python_tag|>{"name": "get_weather", "parameters": {"city": "Paris", "days": 2}}',
    "llama4_json": '{"name": "get_weather", "parameters": {"city": "Paris", "days": 2}}',
    "mistral": f"Let me check.[TOOL_CALLS][{J}]",
    "pythonic": '[get_weather(city="Paris", days=2)]',
    "llama4_pythonic": '<|python_start|>[get_weather(city="Paris", days=2)]<|python_end|>',
    "granite": f"Let me check.<|tool_call|>[{J}]",
    "granite-20b-fc": f"Let me check.<function_call> {J}",
    "internlm": 'Let me check.<|action_start|><|plugin|>{"name": "get_weather", "parameters": {"city": "Paris", '
                '"days": 2}}<|action_end|>',
    "jamba": f"Let me check.<tool_calls>[{J}]</tool_calls>",
    "phi4_mini_json": f"Let me check.functools[{J}]",
    "xlam": f"Let me check.```json\n[{J}]\n```",
    "deepseek_v3": "Let me check.<｜tool▁calls▁begin｜><｜tool▁call▁begin｜>function<｜tool▁sep｜>get_weather\n```json\n"
                   '{"city": "Paris", "days": 2}\n```<｜tool▁call▁end｜><｜tool▁calls▁end｜>',
    "kimi_k2": "Let me check.<|tool_calls_section_begin|><|tool_call_begin|>functions.get_weather:0"
               '<|tool_call_argument_begin|>{"city": "Paris", "days": 2}<|tool_call_end|><|tool_calls_section_end|>',
    "minimax": f"Let me check.<tool_calls>\n{J}\n</tool_calls>",
}


def test_every_registered_tool_parser_has_a_case():
    assert set(CASES) == set(TOOL_PARSERS)


@pytest.mark.parametrize("name", sorted(CASES))
def test_tool_parser_extracts_the_call(name):
    content, calls = get_tool_parser(name).extract(CASES[name])
    assert len(calls) == 1, (name, content)
    assert calls[0].function.name == "get_weather"
    assert json.loads(calls[0].function.arguments) == CALL["arguments"]
    assert content in (None, "Let me check.")
    # plain text is left alone
    content, calls = get_tool_parser(name).extract("The weather is nice.")
    assert calls == [] and content == "The weather is nice."


@pytest.mark.parametrize("name,text,reasoning,answer", [
    ("deepseek_r1", "<think>two plus two</think>4", "two plus two", "4"),
    ("deepseek_r1", "two plus two</think>4", "two plus two", "4"),
    ("qwen3", "<think>hm</think>4", "hm", "4"),
    ("qwen3", "just 4", None, "just 4"),
    ("granite", "Here is my thought process: hm Here is my response: 4", "hm", "4"),
    ("hunyuan_a13b", "<think>\nhm\n</think>\n<answer>\n4\n</answer>", "hm", "4"),
    ("mistral", "[THINK]hm[/THINK]4", "hm", "4"),
])
def test_reasoning_parsers(name, text, reasoning, answer):
    assert name in REASONING_PARSERS
    assert get_reasoning_parser(name).extract(text) == (reasoning, answer)
