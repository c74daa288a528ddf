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
        grammar_to_regex('root ::= "(" root ")" | "x"')          # recursion: handled by the pushdown matcher


class _ByteVocabTokenizer:
    """Every single byte plus multi-character pieces typical of JSON / code vocabularies."""
    eos_token_id = 0
    all_special_ids = [0]

    def __init__(self, extra=()):
        toks = ["<eos>"] + [f"<0x{b:02X}>" for b in range(256)] + list(extra)
        self.vocab = {t: i for i, t in enumerate(toks)}

    def get_vocab(self):
        return self.vocab


JSON_GRAMMAR = r'''
root ::= value
value ::= object | array | string | number | "true" | "false" | "null"
object ::= "{" ws ( string ws ":" ws value ( ws "," ws string ws ":" ws value )* )? ws "}"
array ::= "[" ws ( value ( ws "," ws value )* )? ws "]"
string ::= "\"" [^"\\]* "\""
number ::= "-"? [0-9]+ ("." [0-9]+)?
ws ::= [ \t\n]*
'''


def test_context_free_grammar_matcher():
    """Recursive grammars run on the pushdown matcher: nesting of any depth, left recursion rewritten,
    right recursion at constant stack depth, indirect left recursion refused."""
    from vllm_distributed_amd.structured_output.cfg import CompiledPDA
    extra = ['{"', '"}', '":', '",', '"]', "]]", "}}", "]}", '": ', "true", "null", "[[", ', "', "12", '"a"', "}]}"]
    tok = _ByteVocabTokenizer(extra)
    be = FSMBackend(tok, len(tok.vocab))
    c = CompiledPDA(JSON_GRAMMAR, be.table)
    for text, ok in [('{"a": [1, 2, {"b": null}]}', True), ("[" * 40 + "1" + "]" * 40, True), ("[[1]", False),
                     ('{"a":}', False), ("[1,]", False), ('"x"', True)]:
        assert c.matches(text.encode()) == ok, text
    lark = '''
    ?start: expr
    expr: expr "+" term | expr "-" term | term
    term: term "*" factor | factor
    factor: NUMBER | "(" expr ")"      // parenthesised
    NUMBER: /[0-9]+/
    '''
    c2 = CompiledPDA(lark, be.table)
    for text, ok in [("1+2*3", True), ("(1+2)*3", True), ("((1))", True), ("1+", False), ("(1", False), ("1)", False)]:
        assert c2.matches(text.encode()) == ok, text
    c3 = CompiledPDA('root ::= list\nlist ::= item list | ""\nitem ::= "a" | "(" list ")"', be.table)
    assert c3.matches(b"a" * 2000) and c3.matches(b"a(a(a)a)") and not c3.matches(b"a(a")
    assert max(len(st) for cfg in c3._cfgs for st in cfg) < 8          # tail calls: no stack growth along the list
    with pytest.raises(ValueError, match="left-recursive"):
        CompiledPDA('a ::= b "x" | "y"\nb ::= a "z" | "w"', be.table)

    # token masks: multi-byte tokens that close several levels are allowed exactly when the stack has them
    g = be.compile(_GD(grammar=JSON_GRAMMAR))
    v = tok.vocab
    row = np.zeros((len(v) + 31) // 32, dtype=np.int32)

    def allowed():
        g.fill_bitmask(row)
        bits = np.unpackbits(row.view(np.uint8), bitorder="little")
        return set(np.nonzero(bits)[0].tolist())

    def b(ch):
        return v[f"<0x{ord(ch):02X}>"]
    a0 = allowed()
    assert v["[["] in a0 and v['{"'] in a0 and v["true"] in a0 and v["12"] in a0 and v["]]"] not in a0 and 0 not in a0
    assert g.accept_tokens([v["[["], v["12"]])
    a1 = allowed()
    assert v["]]"] in a1 and b("]") in a1 and b(",") in a1 and v["]}"] not in a1 and v["}]}"] not in a1 and 0 not in a1
    assert g.validate_tokens([v["]]"], 0]) == [v["]]"], 0]
    assert g.accept_tokens([b(","), v['{"'], b("k"), v['": '], v['{"'], b("x"), v['":'], v["[["], b("1")])
    a2 = allowed()                        # [[12,{"k": {"x":[[1   -> "]]" closes both arrays, "]}" does not exist yet
    assert v["]]"] in a2 and v["]}"] not in a2
    assert g.accept_tokens([v["]]"]])
    assert v["}}"] in allowed() and v["}]}"] not in allowed()
    assert g.accept_tokens([v["}}"]]) and v["]]"] in allowed()
    assert g.accept_tokens([v["]]"]])
    assert allowed() == {0}               # complete: only EOS
    assert g.accept_tokens([0]) and g.is_terminated()
    g.rollback(2)
    assert v["]]"] in allowed()


def test_grammar_masks_on_a_large_vocabulary_are_fast():
    """128 K tokens: the first mask of a configuration costs a numpy walk plus the new transitions, repeats
    come from the cache."""
    import itertools
    import time
    alphabet = 'abcdefghijklmnopqrstuvwxyz0123456789 {}[]":,.-_'
    extra = ["".join(p) for p in itertools.islice(itertools.product(alphabet, repeat=3), 100_000)]
    extra += ["".join(p) for p in itertools.islice(itertools.product(alphabet[26:], repeat=4), 28_000)]
    tok = _ByteVocabTokenizer(extra)
    be = FSMBackend(tok, len(tok.vocab))
    g = be.compile(_GD(grammar=JSON_GRAMMAR))
    row = np.zeros((len(tok.vocab) + 31) // 32, dtype=np.int32)
    text = '{"abc": [1, {"k": "v"}, [[2, 3], "x"]], "d": {"e": null}}'
    times = []
    for ch in text.encode():
        t0 = time.perf_counter()
        g.fill_bitmask(row)
        times.append(time.perf_counter() - t0)
        assert g.accept_tokens([1 + ch])
    g.fill_bitmask(row)
    assert row.view(np.uint32)[0] & 1                     # complete: EOS allowed
    assert max(times) < 2.0 and sorted(times)[len(times) // 2] < 0.25, (max(times), sorted(times)[len(times) // 2])
    g2 = be.compile(_GD(grammar=JSON_GRAMMAR))           # a second request reuses every cached mask
    t0 = time.perf_counter()
    for ch in text.encode():
        g2.fill_bitmask(row)
        assert g2.accept_tokens([1 + ch])
    assert time.perf_counter() - t0 < 0.2


def test_structural_tag_lowering():
    from vllm_distributed_amd.structured_output.backends import structural_tag_to_regex
    spec = {"structures": [{"begin": "<function=get_weather>", "end": "</function>",
                            "schema": {"type": "object", "properties": {"city": {"type": "string"}}, "required": ["city"]}},
                           {"begin": "<function=now>", "schema": {"type": "object", "properties": {}}, "end": "</function>"}],
            "triggers": ["<function="]}
    d = compile_regex(structural_tag_to_regex(spec))
    assert d.matches(b'Let me check. <function=get_weather>{"city": "Paris"}</function> Done.')
    assert d.matches(b"no calls at all")
    assert d.matches(b'<function=now>{}</function><function=get_weather>{"city":"x"}</function>')
    assert not d.matches(b'<function=get_weather>{"town": "Paris"}</function>')
    assert not d.matches(b'<function=other>{}</function>')
    assert not d.matches(b'<function=now>{}')
    with pytest.raises(ValueError):
        structural_tag_to_regex({"structures": [{"begin": "[x]", "schema": {}, "end": ""}], "triggers": ["<"]})


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


def test_recursive_grammar_end_to_end(tmp_path):
    """A context-free constraint (w5^n w7 w6^n) through the engine."""
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.sampling_params import GuidedDecodingParams
    path, _ = _save_tiny(str(tmp_path), "llama")
    _add_tokenizer(path)
    llm = LLM(model=path, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8, max_num_batched_tokens=64)
    try:
        gd = GuidedDecodingParams(grammar='root ::= nest\nnest ::= "w5" nest "w6" | "w7"')
        sps = [SamplingParams(temperature=1.0, seed=i, max_tokens=24, guided_decoding=gd,
                              logit_bias={5: 4.0, 7: 3.0, 6: 3.0}) for i in range(4)]
        outs = llm.generate(["w1 w2 w3"] * 4, sps)
        closed = 0
        for o in outs[:4]:
            ids = list(o.outputs[0].token_ids)
            if o.outputs[0].finish_reason == "stop":
                assert ids[-1] == 318
                body = ids[:-1]
                n = body.index(7)
                assert body == [5] * n + [7] + [6] * n, body
                closed += 1
            else:                                   # ran out of tokens: still a prefix of the language
                n = ids.index(7) if 7 in ids else len(ids)
                assert ids[:n] == [5] * n and all(t == 6 for t in ids[n + 1:]) and len(ids) - n - 1 <= n
        assert closed >= 1
    finally:
        llm.shutdown()


def test_structural_tag_response_format_reaches_the_sampling_params():
    from vllm_distributed_amd.entrypoints.openai.protocol import ChatCompletionRequest
    req = ChatCompletionRequest(model="m", messages=[{"role": "user", "content": "hi"}], response_format={
        "type": "structural_tag", "triggers": ["<f="],
        "structures": [{"begin": "<f=a>", "schema": {"type": "object"}, "end": "</f>"}]})
    gd = req._guided()
    assert gd.kind == "structural_tag" and gd.structural_tag["triggers"] == ["<f="]
