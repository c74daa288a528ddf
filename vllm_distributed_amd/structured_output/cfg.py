"""Context-free grammars for guided decoding: a byte-level pushdown automaton, determinised lazily.

A grammar (EBNF / GBNF / Lark-style ``name: ...`` rules) is split into its *recursive* rules (those that can
reach themselves) and the rest.  Non-recursive rules are inlined; every recursive rule becomes a DFA over
bytes plus one "call rule k" symbol per recursive rule (regex_fsm.compile_regex with extra symbols).  A
matcher configuration is a set of stacks of DFA states:

    byte b     top state s -> trans[s, b]
    call k     top state s with an edge (k -> s'): replace s by the return state s', push start(k);
               when s' accepts and has no way on, the return frame is dropped (tail call: right-recursive
               lists run at constant depth)
    return     an accepting top state may pop; the empty stack is the end of the grammar

Each distinct configuration gets an integer id and a row of a transition table ``cfg x byte -> cfg`` that is
filled on demand, so the matcher looks like a DFA to the rest of the backend: the token bitmask of a
configuration is computed for the whole vocabulary with the same column-by-column numpy walk as for regexes
(only the (configuration, byte) pairs not seen before cost Python time) and cached per configuration.
Direct left recursion (``e ::= e "+" t | t``) is rewritten to iteration; indirect left recursion is refused.

Plays the role of the pushdown / Earley matchers inside the reference's grammar backends
(vllm/v1/structured_output/backend_xgrammar.py:95-123 compile_grammar, backend_guidance.py:101-116,
utils.py convert_lark_to_ebnf).
"""
from __future__ import annotations

import re
import threading
from collections import OrderedDict
from typing import Optional

import numpy as np

from .regex_fsm import DFA, compile_regex

_TOKEN = re.compile(r'\s*("(?:[^"\\]|\\.)*"|\'(?:[^\'\\]|\\.)*\'|/(?:[^/\\\n]|\\.)+/|\[(?:[^\]\\]|\\.)*\]|[A-Za-z_][\w\-]*|[()|*+?])')


def _strip_comment(line: str) -> str:
    """Cut a trailing ``# ...`` / ``// ...`` comment that starts between tokens (not inside a terminal)."""
    pos = 0
    while pos < len(line):
        rest = line[pos:]
        lead = len(rest) - len(rest.lstrip())
        if rest[lead:lead + 1] == "#" or rest[lead:lead + 2] == "//":
            return line[:pos].rstrip()
        m = _TOKEN.match(line, pos)
        pos = m.end() if m else pos + lead + 1
    return line


def parse_rules(text: str) -> tuple[dict[str, list[str]], str]:
    """Grammar text -> {rule name: token list}, start rule.  Accepts ``::=`` (GBNF / EBNF) and ``:`` (Lark),
    continuation lines starting with ``|``, ``#`` and ``//`` comments."""
    rules: dict[str, str] = {}
    cur = None
    for raw in text.splitlines():
        line = _strip_comment(raw.rstrip())
        if not line.strip():
            continue
        if line.lstrip().startswith("%"):
            raise ValueError(f"grammar directive {line.strip().split()[0]!r} is not supported")
        m = re.match(r"\s*[?!]?([A-Za-z_][\w\-]*)(?:\.\d+)?\s*(::=|:)\s*(.*)", line)
        if m and not line.lstrip().startswith("|"):
            cur = m.group(1)
            rules[cur] = m.group(3)
        elif cur is not None:
            rules[cur] += " " + line.strip()
    if not rules:
        raise ValueError("empty grammar")
    start = "root" if "root" in rules else ("start" if "start" in rules else next(iter(rules)))
    toks: dict[str, list[str]] = {}
    for name, body in rules.items():
        out, pos = [], 0
        while pos < len(body):
            m = _TOKEN.match(body, pos)
            if not m:
                if body[pos:].strip() == "":
                    break
                raise ValueError(f"cannot parse grammar rule {name!r} near {body[pos:pos + 20]!r}")
            out.append(m.group(1))
            pos = m.end()
        toks[name] = out
    for name, ts in toks.items():
        for t in ts:
            if re.fullmatch(r"[A-Za-z_][\w\-]*", t) and t not in toks:
                raise ValueError(f"undefined grammar rule {t!r}")
    return toks, start


def _refs(tokens: list[str]) -> set[str]:
    return {t for t in tokens if re.fullmatch(r"[A-Za-z_][\w\-]*", t)}


def recursive_rules(rules: dict[str, list[str]]) -> set[str]:
    reach = {n: set(_refs(ts)) for n, ts in rules.items()}
    changed = True
    while changed:
        changed = False
        for n in rules:
            new = set().union(*(reach[m] for m in reach[n])) - reach[n] if reach[n] else set()
            if new:
                reach[n] |= new
                changed = True
    return {n for n in rules if n in reach[n]}


def _split_alternatives(tokens: list[str]) -> list[list[str]]:
    out, cur, depth = [], [], 0
    for t in tokens:
        if t == "(":
            depth += 1
        elif t == ")":
            depth -= 1
        if t == "|" and depth == 0:
            out.append(cur)
            cur = []
        else:
            cur.append(t)
    out.append(cur)
    return out


def _remove_direct_left_recursion(name: str, tokens: list[str]) -> list[str]:
    """``A ::= A x | A y | b | c``  ->  ``A ::= (b | c) (x | y)*``."""
    alts = _split_alternatives(tokens)
    left = [a[1:] for a in alts if a and a[0] == name]
    if not left:
        return tokens
    base = [a for a in alts if not (a and a[0] == name)]
    if not base or any(not a for a in left):
        raise ValueError(f"grammar rule {name!r} is left-recursive without a base case")

    def group(branches):
        out = ["("]
        for i, b in enumerate(branches):
            if i:
                out.append("|")
            out += ["("] + b + [")"] if b else ["(", ")"]
        return out + [")"]
    return group(base) + group(left) + ["*"]


def _literal(tok: str) -> str:
    """Quoted terminal -> regex for exactly that text."""
    def unescape(m):
        c = m.group(1)
        if c[0] == "x":
            return chr(int(c[1:], 16))
        if c[0] == "u":
            return chr(int(c[1:], 16))
        return {"n": "\n", "t": "\t", "r": "\r", "0": "\0"}.get(c, c)
    return re.escape(re.sub(r"\\(x[0-9A-Fa-f]{2}|u[0-9A-Fa-f]{4}|.)", unescape, tok[1:-1]))


def lower_rule(rules: dict[str, list[str]], name: str, rule_index: dict[str, int], stack: tuple = ()) -> str:
    """Regex of rule ``name``: rules in ``rule_index`` become call symbols, every other rule is inlined."""
    out = []
    for t in rules[name]:
        if t[0] in "\"'":
            out.append(_literal(t))
        elif t[0] == "/":
            out.append("(" + t[1:-1] + ")")
        elif t[0] == "[" or t in ("(", ")", "|", "*", "+", "?"):
            out.append(t)
        elif t in rule_index:
            out.append("\\R{%d}" % rule_index[t])
        else:
            if t in stack:
                raise ValueError(f"recursive grammar rule {t!r}")
            out.append("(" + lower_rule(rules, t, rule_index, stack + (t,)) + ")")
    return "".join(out)


def grammar_to_regex(text: str) -> str:
    """A grammar without recursion is a regular expression: every rule inlined into the start rule."""
    rules, start = parse_rules(text)
    if recursive_rules(rules):
        raise ValueError("recursive grammar")
    return lower_rule(rules, start, {}, (start,))


class CompiledPDA:
    """A grammar bound to a vocabulary.  Exposes what Grammar (backends.py) uses of CompiledGrammar:
    ``dfa.start / dfa.dead / dfa.accepting[s]``, ``advance``, ``mask_for``, ``table``."""

    MAX_DEPTH = 256
    MAX_CONFIGS = 1 << 20

    class _View:
        def __init__(self, owner):
            self.o = owner
            self.start, self.dead = 1, 0

        class _Acc:
            def __init__(self, o):
                self.o = o

            def __getitem__(self, s):
                return bool(self.o._accepting[int(s)])

        @property
        def accepting(self):
            return CompiledPDA._View._Acc(self.o)

    def __init__(self, grammar_text: str, table):
        self.table = table
        self.words = (table.vocab_size + 31) // 32
        rules, start = parse_rules(grammar_text)
        rules = {n: _remove_direct_left_recursion(n, ts) for n, ts in rules.items()}
        rec = sorted(recursive_rules(rules))
        if start not in rec:
            rec.append(start)                           # the start rule always gets an automaton of its own
        self.rule_index = {n: i for i, n in enumerate(rec)}
        R = len(rec)

        dfas: list[DFA] = [compile_regex(lower_rule(rules, n, self.rule_index), max_states=50_000, extra_symbols=R)
                           for n in rec]
        # one global numbering of the live states of all rule automata
        offs, total = [], 0
        for d in dfas:
            offs.append(total)
            total += d.num_states
        T = np.full((total, 256), -1, dtype=np.int32)
        self._calls: list[list[tuple[int, int]]] = [[] for _ in range(total)]
        self._acc = np.zeros(total, dtype=bool)
        for r, d in enumerate(dfas):
            tr = d.trans
            for s in range(d.num_states):
                if s == d.dead:
                    continue
                g = offs[r] + s
                row = tr[s, :256]
                T[g] = np.where(row == d.dead, -1, row + offs[r])
                self._acc[g] = d.accepting[s]
                for k in range(R):
                    s2 = int(tr[s, 256 + k])
                    if s2 != d.dead:
                        self._calls[g].append((k, offs[r] + s2))
        self._T = T
        self._has_bytes = (T >= 0).any(axis=1)
        self._starts = [offs[r] + d.start for r, d in enumerate(dfas)]
        self._pure_final = np.array([bool(self._acc[g]) and not self._has_bytes[g] and not self._calls[g]
                                     for g in range(total)])
        self._check_left_recursion(rec)
        # lazily determinised configurations: id 0 = dead, id 1 = start
        self._lock = threading.RLock()
        self._closure_memo: dict[tuple, frozenset] = {}
        self._cfg_ids: dict[frozenset, int] = {frozenset(): 0}
        self._cfgs: list[frozenset] = [frozenset()]
        self._ctrans = np.full((1024, 256), -1, dtype=np.int32)
        self._ctrans[0, :] = 0
        self._accepting = np.zeros(1024, dtype=bool)
        s0 = self._intern(self._closure((self._starts[self.rule_index[start]],)))
        assert s0 == 1
        self._masks: "OrderedDict[int, np.ndarray]" = OrderedDict()
        self.dfa = CompiledPDA._View(self)

    # ---- static checks -----------------------------------------------------------------------------------
    def _check_left_recursion(self, names: list[str]) -> None:
        """Rule k is a left corner of rule r if r can call k before consuming a byte."""
        R = len(names)
        nullable = [False] * R
        changed = True

        def first_calls(r: int) -> tuple[set[int], bool]:
            seen, todo, calls, null = set(), [self._starts[r]], set(), False
            while todo:
                g = todo.pop()
                if g in seen:
                    continue
                seen.add(g)
                null |= bool(self._acc[g])
                for k, ret in self._calls[g]:
                    calls.add(k)
                    if nullable[k]:
                        todo.append(ret)
            return calls, null
        while changed:
            changed = False
            for r in range(R):
                _, null = first_calls(r)
                if null and not nullable[r]:
                    nullable[r] = changed = True
        left = [first_calls(r)[0] for r in range(R)]
        for r in range(R):
            seen, todo = set(), list(left[r])
            while todo:
                k = todo.pop()
                if k == r:
                    raise ValueError(f"grammar rule {names[r]!r} is left-recursive through other rules; rewrite it "
                                     "with repetition (x*) instead")
                if k not in seen:
                    seen.add(k)
                    todo.extend(left[k])

    # ---- configurations ----------------------------------------------------------------------------------
    def _closure(self, stack: tuple) -> frozenset:
        memo = self._closure_memo.get(stack)
        if memo is not None:
            return memo
        out, seen, todo = set(), set(), [stack]
        while todo:
            st = todo.pop()
            if st in seen:
                continue
            seen.add(st)
            if not st:
                out.add(st)
                continue
            g = st[-1]
            if self._has_bytes[g]:
                out.add(st)
            if self._acc[g]:
                todo.append(st[:-1])
            for k, ret in self._calls[g]:
                base = st[:-1] if self._pure_final[ret] else st[:-1] + (ret,)
                if len(base) < self.MAX_DEPTH:
                    todo.append(base + (self._starts[k],))
        res = frozenset(out)
        if len(self._closure_memo) > 200_000:
            self._closure_memo.clear()
        self._closure_memo[stack] = res
        return res

    def _intern(self, cfg: frozenset) -> int:
        i = self._cfg_ids.get(cfg)
        if i is not None:
            return i
        i = len(self._cfgs)
        if i >= self.MAX_CONFIGS:
            return 0
        if i >= self._ctrans.shape[0]:
            grow = self._ctrans.shape[0]
            self._ctrans = np.vstack([self._ctrans, np.full((grow, 256), -1, dtype=np.int32)])
            self._accepting = np.concatenate([self._accepting, np.zeros(grow, dtype=bool)])
        self._cfg_ids[cfg] = i
        self._cfgs.append(cfg)
        self._accepting[i] = () in cfg
        return i

    def _step(self, cfg: int, b: int) -> int:
        nxt = int(self._ctrans[cfg, b])
        if nxt >= 0:
            return nxt
        out: set = set()
        T = self._T
        for st in self._cfgs[cfg]:
            if st:
                g2 = int(T[st[-1], b])
                if g2 >= 0:
                    out |= self._closure(st[:-1] + (g2,))
        nxt = self._intern(frozenset(out))
        self._ctrans[cfg, b] = nxt
        return nxt

    # ---- CompiledGrammar interface -----------------------------------------------------------------------
    def advance(self, state: int, token: int) -> int:
        if token < 0 or token >= self.table.vocab_size:
            return 0
        data = self.table.tokens[token]
        if not data:
            return 0
        with self._lock:
            for b in data:
                state = self._step(state, b)
                if state == 0:
                    return 0
        return state

    def _walk(self, state: int) -> np.ndarray:
        t = self.table
        cur = np.full(t.vocab_size, state, dtype=np.int32)
        for j in range(t.max_len):
            idx = t.order[: t.count_ge[j]]
            if idx.size == 0:
                break
            c, b = cur[idx], t.mat[idx, j]
            nxt = self._ctrans[c, b]
            miss = nxt < 0
            if miss.any():
                for key in np.unique(c[miss].astype(np.int64) * 256 + b[miss]):
                    self._step(int(key) >> 8, int(key) & 255)
                nxt = self._ctrans[c, b]
            cur[idx] = nxt
        cur[t.lens == 0] = 0
        return cur

    def mask_for(self, state: int) -> np.ndarray:
        with self._lock:
            m = self._masks.get(state)
            if m is not None:
                self._masks.move_to_end(state)
                return m
            ok = self._walk(state) != 0
            bits = np.zeros(self.words * 32, dtype=np.uint8)
            bits[: self.table.vocab_size] = ok
            m = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).view(np.int32).reshape(-1)
            self._masks[state] = m
            if len(self._masks) > 4096:
                self._masks.popitem(last=False)
            return m

    def matches(self, data: bytes) -> bool:
        """Whole-string membership (tests, validation)."""
        s = 1
        with self._lock:
            for b in data:
                s = self._step(s, b)
                if s == 0:
                    return False
        return bool(self._accepting[s])


def is_recursive_grammar(text: str) -> bool:
    rules, _ = parse_rules(text)
    return bool(recursive_rules(rules))
