"""Guided-decoding backend: every constraint kind is lowered to a regular expression, compiled to
a byte-level DFA (regex_fsm.py) and applied to the vocabulary as a per-state token bitmask.

The bitmask of a DFA state is computed for the whole vocabulary at once: token byte strings sit in
one padded [V, L] matrix and the transition table is applied column by column with numpy gathers
(L steps instead of V * L), then cached per state; compiled grammars are cached per spec.
Plays the role of the reference's xgrammar / guidance / outlines backends
(vllm/v1/structured_output/backend_*.py, backend_types.py).
"""
from __future__ import annotations

import json
import re
import threading
from collections import OrderedDict
from typing import Optional

import numpy as np

from ..logger import init_logger
from .cfg import CompiledPDA, grammar_to_regex, is_recursive_grammar
from .json_schema import any_json, schema_to_regex
from .regex_fsm import DFA, compile_regex

logger = init_logger(__name__)


def _bytes_to_unicode() -> dict[str, int]:
    """Inverse of the GPT-2 byte <-> printable-character table used by byte-level BPE vocabularies."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {chr(c): b for b, c in zip(bs, cs)}


class TokenTable:
    """Byte strings of every token id, as a padded matrix."""

    def __init__(self, tokenizer, vocab_size: int):
        self.vocab_size = vocab_size
        vocab = tokenizer.get_vocab()
        special = set(getattr(tokenizer, "all_special_ids", []) or [])
        byte_level = sum(1 for t in list(vocab)[:2000] if t.startswith("Ġ")) > 5
        inv = _bytes_to_unicode() if byte_level else None
        toks: list[bytes] = [b""] * vocab_size
        for tok, i in vocab.items():
            if i >= vocab_size or i in special:
                continue
            if byte_level:
                try:
                    toks[i] = bytes(inv[c] for c in tok)
                except KeyError:
                    toks[i] = tok.encode("utf-8")
            elif re.fullmatch(r"<0x[0-9A-Fa-f]{2}>", tok):
                toks[i] = bytes([int(tok[3:5], 16)])
            elif "▁" in tok:
                toks[i] = tok.replace("▁", " ").encode("utf-8")
            else:
                toks[i] = tok.encode("utf-8")
        self.tokens = toks
        self.eos_id = getattr(tokenizer, "eos_token_id", None)
        L = max((len(t) for t in toks), default=1)
        self.max_len = L
        self.mat = np.zeros((vocab_size, L), dtype=np.uint8)
        self.lens = np.zeros(vocab_size, dtype=np.int32)
        for i, t in enumerate(toks):
            self.lens[i] = len(t)
            if t:
                self.mat[i, :len(t)] = np.frombuffer(t, dtype=np.uint8)
        # tokens sorted by length let each column touch only the tokens that are long enough
        self.order = np.argsort(-self.lens, kind="stable")
        self.count_ge = [int((self.lens > j).sum()) for j in range(L)]


class CompiledGrammar:
    """A DFA bound to a vocabulary; shared by every request with the same constraint."""

    def __init__(self, dfa: DFA, table: TokenTable):
        self.dfa, self.table = dfa, table
        self.words = (table.vocab_size + 31) // 32
        self._masks: "OrderedDict[int, np.ndarray]" = OrderedDict()
        self._lock = threading.Lock()
        self.live = self._token_liveness()

    def _walk(self, state: int) -> np.ndarray:
        """DFA state reached from ``state`` by every token of the vocabulary (int32 [V])."""
        t, d = self.table, self.dfa
        cur = np.full(t.vocab_size, state, dtype=np.int32)
        for j in range(t.max_len):
            idx = t.order[: t.count_ge[j]]
            if idx.size == 0:
                break
            cur[idx] = d.trans[cur[idx], t.mat[idx, j]]
        cur[t.lens == 0] = d.dead
        return cur

    def _token_liveness(self, max_states: int = 3000) -> np.ndarray:
        """live[s]: some sequence of whole *tokens* leads from s to an accepting state.  A byte-level
        prefix can be viable for the DFA yet have no token that continues it; such states are pruned
        so that sampling never paints itself into a corner."""
        d = self.dfa
        live = np.ones(d.num_states, dtype=bool)
        live[d.dead] = False
        if d.num_states > max_states:
            return live
        succ: dict[int, np.ndarray] = {}
        todo = [d.start]
        while todo:
            s = todo.pop()
            if s in succ:
                continue
            nxt = np.unique(self._walk(s))
            nxt = nxt[nxt != d.dead]
            succ[s] = nxt
            todo.extend(int(x) for x in nxt if int(x) not in succ)
        ok = {s for s in succ if d.accepting[s]}
        changed = True
        while changed:
            changed = False
            for s, nxt in succ.items():
                if s not in ok and any(int(x) in ok for x in nxt):
                    ok.add(s)
                    changed = True
        for s in succ:
            if s not in ok:
                live[s] = False
        return live

    def mask_for(self, state: int) -> np.ndarray:
        with self._lock:
            m = self._masks.get(state)
            if m is not None:
                self._masks.move_to_end(state)
                return m
        t = self.table
        ok = self.live[self._walk(state)]
        bits = np.zeros(self.words * 32, dtype=np.uint8)
        bits[: t.vocab_size] = ok
        m = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).view(np.int32).reshape(-1)
        with self._lock:
            self._masks[state] = m
            if len(self._masks) > 4096:
                self._masks.popitem(last=False)
        return m

    def advance(self, state: int, token: int) -> int:
        if token < 0 or token >= self.table.vocab_size:
            return self.dfa.dead
        data = self.table.tokens[token]
        if not data:
            return self.dfa.dead
        tr = self.dfa.trans
        for b in data:
            state = int(tr[state, b])
        return state if self.live[state] else self.dfa.dead


class Grammar:
    """Per-request matcher state."""

    def __init__(self, compiled: CompiledGrammar):
        self.c = compiled
        self.state = compiled.dfa.start
        self.history: list[int] = []
        self.terminated = False

    def _eos(self) -> Optional[int]:
        return self.c.table.eos_id

    def accept_tokens(self, tokens: list[int]) -> bool:
        for t in tokens:
            if self.terminated:
                return False
            self.history.append(self.state)
            if t == self._eos():
                if not self.c.dfa.accepting[self.state]:
                    self.history.pop()
                    return False
                self.terminated = True
                continue
            nxt = self.c.advance(self.state, t)
            if nxt == self.c.dfa.dead:
                self.history.pop()
                return False
            self.state = nxt
        return True

    def validate_tokens(self, tokens: list[int]) -> list[int]:
        """Longest prefix of ``tokens`` the grammar accepts; the matcher state is unchanged."""
        s, out = self.state, []
        for t in tokens:
            if t == self._eos():
                if self.c.dfa.accepting[s]:
                    out.append(t)
                break
            s = self.c.advance(s, t)
            if s == self.c.dfa.dead:
                break
            out.append(t)
        return out

    def rollback(self, n: int) -> None:
        for _ in range(min(n, len(self.history))):
            self.state = self.history.pop()
            self.terminated = False

    def fill_bitmask(self, row: np.ndarray) -> None:
        eos = self._eos()
        if self.terminated:
            row[:] = 0
        else:
            row[:] = self.c.mask_for(self.state)
            if not self.c.dfa.accepting[self.state]:
                eos = None
            # a state without outgoing tokens must be able to stop
        if eos is not None and 0 <= eos < self.c.table.vocab_size:
            row.view(np.uint32)[eos // 32] |= np.uint32(1 << (eos % 32))

    def is_terminated(self) -> bool:
        return self.terminated

    def reset(self) -> None:
        self.state, self.history, self.terminated = self.c.dfa.start, [], False


# ---- constraint front ends ------------------------------------------------------------------------
def choice_to_regex(choices: list[str]) -> str:
    return "(" + "|".join(re.escape(c) for c in choices) + ")"


def structural_tag_to_regex(spec, any_ws: bool = True) -> str:
    """``{"structures": [{"begin", "schema", "end"}...], "triggers": [...]}``: free text, and whenever a
    trigger string starts, one of the structures whose ``begin`` starts with it: begin, a JSON value of its
    schema, end; then free text again.  Free text may not contain the first character of any trigger (a
    regular over-approximation of "does not contain a trigger" that keeps the automaton small).
    Reference: vllm/v1/structured_output/backend_xgrammar.py:107-123."""
    if isinstance(spec, str):
        spec = json.loads(spec)
    structures, triggers = spec.get("structures") or [], spec.get("triggers") or []
    if not structures or not triggers:
        raise ValueError("a structural tag needs 'structures' and 'triggers'")
    firsts = sorted({t[0] for t in triggers if t})
    if any(ord(c) > 127 for c in firsts):
        raise ValueError("structural tag triggers must start with an ASCII character")
    free = "[^" + "".join("\\" + c if c in "\\]^-" else c for c in firsts) + "]*"
    tagged = []
    for st in structures:
        if not any(st["begin"].startswith(t) for t in triggers):
            raise ValueError(f"structure {st['begin']!r} starts with none of the triggers")
        body = schema_to_regex(st.get("schema") or {}, any_ws)
        tagged.append(re.escape(st["begin"]) + "(" + body + ")" + re.escape(st["end"]))
    return free + "((" + "|".join(tagged) + ")" + free + ")*"


class FSMBackend:
    def __init__(self, tokenizer, vocab_size: int, disable_any_whitespace: bool = False):
        self.table = TokenTable(tokenizer, vocab_size)
        self.any_ws = not disable_any_whitespace
        self._cache: "OrderedDict[tuple, CompiledGrammar]" = OrderedDict()
        self._lock = threading.Lock()

    def to_regex(self, gd) -> tuple[str, str]:
        if gd.json is not None:
            schema = gd.json if isinstance(gd.json, (dict, str)) else gd.json
            return "json", schema_to_regex(schema, self.any_ws)
        if getattr(gd, "json_object", None):
            ws = "[ ]?" if self.any_ws else ""
            obj = any_json(3, ws)
            return "json_object", rf"\{{{ws}(\"([^\"\\\x00-\x1f]|\\.)*\"{ws}:{ws}{obj}({ws},{ws}\"([^\"\\\x00-\x1f]|\\.)*\"{ws}:{ws}{obj})*)?{ws}\}}"
        if gd.regex is not None:
            return "regex", gd.regex
        if gd.choice is not None:
            return "choice", choice_to_regex(list(gd.choice))
        if gd.grammar is not None:
            return "grammar", grammar_to_regex(gd.grammar)
        if getattr(gd, "structural_tag", None):
            return "structural_tag", structural_tag_to_regex(gd.structural_tag, self.any_ws)
        raise ValueError("empty guided-decoding request")

    def compile(self, gd) -> Grammar:
        if gd.grammar is not None and gd.json is None and is_recursive_grammar(gd.grammar):
            kind, rx = "cfg", gd.grammar                 # pushdown matcher (cfg.py)
        else:
            kind, rx = self.to_regex(gd)
        key = (kind, rx)
        with self._lock:
            c = self._cache.get(key)
            if c is not None:
                self._cache.move_to_end(key)
        if c is None:
            c = CompiledPDA(rx, self.table) if kind == "cfg" else CompiledGrammar(compile_regex(rx), self.table)
            with self._lock:
                self._cache[key] = c
                if len(self._cache) > 128:
                    self._cache.popitem(last=False)
        return Grammar(c)

    def validate(self, gd) -> None:
        """Raise ValueError for constraints this backend cannot enforce (called at request intake)."""
        if gd.grammar is not None and gd.json is None and is_recursive_grammar(gd.grammar):
            CompiledPDA(gd.grammar, self.table)
            return
        kind, rx = self.to_regex(gd)
        compile_regex(rx, max_states=50_000)
