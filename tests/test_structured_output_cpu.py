"""Guided decoding: regex / JSON-schema / grammar lowering, token bitmasks, and end-to-end
generation that must satisfy the constraint (modelled on tests/v1/structured_output and
tests/v1/entrypoints/llm/test_struct_output_generate.py)."""
import json
import re

import numpy as np
import pytest

from tests.test_e2e_cpu import _save_tiny
from tests.test_openai_server_cpu import _add_tokenizer
from vllm_distributed_amd.structured_output.backends import FSMBackend, grammar_to_regex
from vllm_distributed_amd.structured_output.json_schema import schema_to_regex
from vllm_distributed_amd.structured_output.regex_fsm import compile_regex


@pytest.mark.parametrize("pat,good,bad", [
    (r"\d{3}-\d{4}", ["555-1234"], ["5555-123", "555-12345"]),
    (r"(yes|no|maybe)", ["yes", "maybe"], ["ye", "nope"]),
    (r'"[^"\\]*(\\.[^"\\]*)*"', ['"abc"', '"a\\"b"', '"é漢"'], ['"abc', 'abc"']),
    (r"-?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)?", ["0", "-12.5e+3"], ["01", "1.", "--1"]),
    (r".{2,4}", ["ab", "漢字ab"], ["a", "abcde"]),
    (r"[^a-c]+x", ["ddx", "éx"], ["ax", "x"]),
])
def test_regex_dfa_agrees_with_re(pat, good, bad):
    d = compile_regex(pat)
    for g in good:
        assert re.fullmatch(pat, g) and d.matches(g.encode())
    for b in bad:
        assert not re.fullmatch(pat, b) and not d.matches(b.encode())


def test_json_schema_regex():
    schema = {"type": "object",
              "properties": {"name": {"type": "string"}, "age": {"type": "integer"},
                             "tags": {"type": "array", "items": {"type": "string"}, "maxItems": 3},
                             "kind": {"enum": ["a", "b"]}, "addr": {"$ref": "#/$defs/addr"}},
              "required": ["name", "age"],
              "$defs": {"addr": {"type": "object", "properties": {"zip": {"type": "string", "pattern": "[0-9]{5}"}},
                                 "required": ["zip"]}}}
    d = compile_regex(schema_to_regex(schema))
    for g in [{"name": "x", "age": 3}, {"name": "é", "age": -10, "tags": ["a", "b"], "kind": "b", "addr": {"zip": "12345"}}]:
        assert d.matches(json.dumps(g, ensure_ascii=False).encode())
        assert d.matches(json.dumps(g, ensure_ascii=False, separators=(",", ":")).encode())
    for b in ['{"age": 3}', '{"name": "x", "age": 3.5}', '{"name": "x", "age": 3, "tags": ["a","b","c","d"]}',
              '{"name": "x", "age": 3, "addr": {"zip": "1234"}}']:
        assert not d.matches(b.encode())
    with pytest.raises(ValueError):
        schema_to_regex({"$ref": "#/$defs/n", "$defs": {"n": {"type": "object", "properties": {"next": {"$ref": "#/$defs/n"}},
                                                            "required": ["next"]}}})


def test_grammar_lowering():
    rx = grammar_to_regex('root ::= "SELECT " col " FROM " tbl\ncol ::= [a-z]+ | "*"\ntbl ::= "users" | "orders"')
    d = compile_regex(rx)
    assert d.matches(b"SELECT * FROM users") and d.matches(b"SELECT id FROM orders")
    assert not d.matches(b"SELECT * FROM x")
    with pytest.raises(ValueError):
        grammar_to_regex('root ::= "(" root ")" | "x"')


class _GD:
    def __init__(self, **kw):
        self.json = self.regex = self.choice = self.grammar = None
        self.json_object = None
        self.__dict__.update(kw)


def test_token_bitmask_and_matcher(tmp_path):
    from vllm_distributed_amd.engine.processor import get_tokenizer
    path, _ = _save_tiny(str(tmp_path), "llama")
    _add_tokenizer(path)
    tok = get_tokenizer(path)
    be = FSMBackend(tok, 320)
    g = be.compile(_GD(regex=r"w1(w22|w3)+"))
    row = np.zeros(10, dtype=np.int32)

    def allowed():
        g.fill_bitmask(row)
        bits = np.unpackbits(row.view(np.uint8), bitorder="little")
        return set(np.nonzero(bits)[0].tolist())
    # token "w1" and every longer token that is still a viable prefix ("w1" only: w1x would need x in (w22|w3))
    assert allowed() == {1}
    assert g.accept_tokens([1])
    # "w2" is a byte prefix of "w22" but no token continues it ("2" is not a token): pruned
    assert allowed() == {22, 3}
    assert g.validate_tokens([3, 22, 5]) == [3, 22]
    assert g.accept_tokens([3]) and allowed() == {22, 3, tok.eos_token_id}
    assert not g.accept_tokens([7])
    assert g.accept_tokens([22, tok.eos_token_id]) and g.is_terminated()
    g.rollback(2)
    assert not g.is_terminated() and allowed() == {22, 3, tok.eos_token_id}


def test_guided_generation_end_to_end(tmp_path):
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.sampling_params import GuidedDecodingParams
    path, _ = _save_tiny(str(tmp_path), "llama")
    _add_tokenizer(path)
    llm = LLM(model=path, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8, max_num_batched_tokens=64)
    try:
        sps = [SamplingParams(temperature=0.8, seed=i, max_tokens=12,
                              guided_decoding=GuidedDecodingParams(regex=r"w7(w8|w9){2,3}w10")) for i in range(3)]
        sps.append(SamplingParams(temperature=0, max_tokens=12,
                                  guided_decoding=GuidedDecodingParams(choice=["w100w101", "w200"])))
        sps.append(SamplingParams(temperature=0, max_tokens=6))       # unconstrained rides along
        outs = llm.generate(["w1 w2 w3"] * 5, sps)
        for o in outs[:3]:
            ids = list(o.outputs[0].token_ids)
            assert ids[-1] == 318 and o.outputs[0].finish_reason == "stop"      # EOS once the pattern is complete
            s = "".join(f"w{t}" for t in ids[:-1])
            assert re.fullmatch(r"w7(w8|w9){2,3}w10", s), s
        ids = list(outs[3].outputs[0].token_ids)
        assert ids in ([100, 101, 318], [200, 318])
        assert len(outs[4].outputs[0].token_ids) == 6
    finally:
        llm.shutdown()
