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


def grammar_to_regex(text: str) -> str:
    """Non-recursive EBNF / GBNF (``name ::= ...`` rules, "literals", [classes], ( ) | * + ?) -> regex.
    The start rule is ``root`` (or ``start``, or the first rule)."""
    rules: dict[str, str] = {}
    cur = None
    for line in text.splitlines():
        line = line.split("#", 1)[0].rstrip() if not re.search(r'"[^"]*#', line) else line.rstrip()
        if not line.strip():
            continue
        m = re.match(r"\s*([A-Za-z_][\w\-]*)\s*(::=|:)\s*(.*)", line)
        if m and not line.lstrip().startswith("|"):
            cur = m.group(1)
            rules[cur] = m.group(3)
        elif cur is not None:
            rules[cur] += " " + line.strip()
    if not rules:
        raise ValueError("empty grammar")
    start = "root" if "root" in rules else ("start" if "start" in rules else next(iter(rules)))
    tok = re.compile(r'\s*("(?:[^"\\]|\\.)*"|\[(?:[^\]\\]|\\.)*\]|[A-Za-z_][\w\-]*|[()|*+?])')

    def expand(name: str, stack: tuple) -> str:
        if name in stack:
            raise ValueError(f"recursive grammar rule {name!r} cannot be lowered to a finite automaton; "
                             "use a JSON schema or a regex")
        if name not in rules:
            raise ValueError(f"undefined grammar rule {name!r}")
        out, pos, body = [], 0, rules[name]
        while pos < len(body):
            m = tok.match(body, pos)
            if not m:
                if body[pos:].strip() == "":
                    break
                raise ValueError(f"cannot parse grammar rule {name!r} near {body[pos:pos + 20]!r}")
            t = m.group(1)
            pos = m.end()
            if t[0] == '"':
                out.append(re.escape(json.loads(t)))
            elif t[0] == "[":
                out.append(t)
            elif t in "()|*+?":
                out.append(t)
            else:
                out.append("(" + expand(t, stack + (name,)) + ")")
        return "".join(out)

    return expand(start, ())


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
            raise ValueError("structural tags are not supported by the built-in guided-decoding backend")
        raise ValueError("empty guided-decoding request")

    def compile(self, gd) -> Grammar:
        kind, rx = self.to_regex(gd)
        key = (kind, rx)
        with self._lock:
            c = self._cache.get(key)
            if c is not None:
                self._cache.move_to_end(key)
        if c is None:
            c = CompiledGrammar(compile_regex(rx), self.table)
            with self._lock:
                self._cache[key] = c
                if len(self._cache) > 128:
                    self._cache.popitem(last=False)
        return Grammar(c)

    def validate(self, gd) -> None:
        """Raise ValueError for constraints this backend cannot enforce (called at request intake)."""
        kind, rx = self.to_regex(gd)
        compile_regex(rx, max_states=50_000)
