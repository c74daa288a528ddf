"""Regular expression -> byte-level DFA, built in-tree (no external grammar library).

Supported syntax: literals, escapes (\\d \\w \\s \\D \\W \\S \\n \\t \\r \\f \\v \\xHH \\uHHHH and escaped
punctuation), ``.``, classes ``[a-z0-9_]`` / ``[^...]``, groups ``( )`` / ``(?: )``, alternation,
quantifiers ``* + ? {m} {m,} {m,n}`` (lazy suffix ``?`` accepted and ignored: matching is a
membership test, greediness has no meaning).  The automaton runs over UTF-8 bytes, so it lines
up with token byte strings; ``.`` and negated classes accept any well-formed multi-byte character.

For context-free grammars (cfg.py) the alphabet can be widened: ``\\R{k}`` in a pattern stands for symbol
256 + k ("call rule k"), and ``compile_regex(..., extra_symbols=n)`` builds the automaton over 256 + n symbols.
"""
from __future__ import annotations

import numpy as np

ALL = (1 << 256) - 1


def _mask(chars) -> int:
    m = 0
    for c in chars:
        m |= 1 << (c if isinstance(c, int) else ord(c))
    return m


def _range(lo: int, hi: int) -> int:
    return ((1 << (hi + 1)) - 1) & ~((1 << lo) - 1)


DIGIT = _range(48, 57)
WORD = DIGIT | _range(65, 90) | _range(97, 122) | _mask("_")
SPACE = _mask(" \t\n\r\f\v")
ASCII = _range(0, 127)


# ---- AST -------------------------------------------------------------------------------------
# ("lit", mask) one byte from mask | ("cat", [..]) | ("alt", [..]) | ("rep", node, lo, hi|None) | ("utf8",) any
# multi-byte char | ("eps",)
class _Parser:
    def __init__(self, pattern: str):
        self.p, self.i = pattern, 0

    def parse(self):
        node = self._alt()
        if self.i != len(self.p):
            raise ValueError(f"unbalanced ')' at {self.i} in regex")
        return node

    def _peek(self):
        return self.p[self.i] if self.i < len(self.p) else None

    def _alt(self):
        branches = [self._cat()]
        while self._peek() == "|":
            self.i += 1
            branches.append(self._cat())
        return branches[0] if len(branches) == 1 else ("alt", branches)

    def _cat(self):
        items = []
        while self._peek() is not None and self._peek() not in "|)":
            items.append(self._quant())
        if not items:
            return ("eps",)
        return items[0] if len(items) == 1 else ("cat", items)

    def _quant(self):
        atom = self._atom()
        while True:
            c = self._peek()
            if c == "*":
                lo, hi = 0, None
            elif c == "+":
                lo, hi = 1, None
            elif c == "?":
                lo, hi = 0, 1
            elif c == "{":
                j = self.p.find("}", self.i)
                body = self.p[self.i + 1:j] if j > 0 else ""
                parts = body.split(",")
                if j < 0 or not all(x.strip().isdigit() or x.strip() == "" for x in parts) or not parts[0].strip():
                    return atom             # a literal brace
                lo = int(parts[0])
                hi = lo if len(parts) == 1 else (int(parts[1]) if parts[1].strip() else None)
                self.i = j
            else:
                return atom
            self.i += 1
            if self._peek() in ("?", "+"):       # lazy / possessive markers
                self.i += 1
            atom = ("rep", atom, lo, hi)

    def _atom(self):
        c = self.p[self.i]
        self.i += 1
        if c == "(":
            if self.p.startswith("?:", self.i):
                self.i += 2
            elif self.p.startswith("?P<", self.i):
                self.i = self.p.index(">", self.i) + 1
            elif self._peek() == "?":
                raise ValueError("look-around and inline flags are not supported in guided regexes")
            node = self._alt()
            if self._peek() != ")":
                raise ValueError("missing ')' in regex")
            self.i += 1
            return node
        if c == "[":
            return self._class()
        if c == ".":
            return ("alt", [("lit", ASCII & ~_mask("\n")), ("utf8",)])
        if c in "^$":
            return ("eps",)                   # whole-string match is implied
        if c == "\\":
            return self._escape(in_class=False)
        return self._char(c)

    @staticmethod
    def _char(c: str):
        b = c.encode("utf-8")
        if len(b) == 1:
            return ("lit", 1 << b[0])
        return ("cat", [("lit", 1 << x) for x in b])

    def _escape(self, in_class: bool):
        c = self.p[self.i]
        self.i += 1
        table = {"d": DIGIT, "w": WORD, "s": SPACE, "D": ASCII & ~DIGIT, "W": ASCII & ~WORD, "S": ASCII & ~SPACE}
        if c in table:
            m = table[c]
            if in_class:
                return m, c in "DWS"
            return ("alt", [("lit", m), ("utf8",)]) if c in "DWS" else ("lit", m)
        simple = {"n": "\n", "t": "\t", "r": "\r", "f": "\f", "v": "\v", "0": "\0", "a": "\a", "b": "\b" if in_class else None}
        if c in simple and simple[c] is not None:
            ch = simple[c]
        elif c == "x":
            ch = chr(int(self.p[self.i:self.i + 2], 16))
            self.i += 2
        elif c == "u":
            ch = chr(int(self.p[self.i:self.i + 4], 16))
            self.i += 4
        elif c == "R" and not in_class and self._peek() == "{":
            j = self.p.index("}", self.i)
            k = int(self.p[self.i + 1:j])
            self.i = j + 1
            return ("lit", 1 << (256 + k))
        elif c == "b" or c == "B":
            return ("eps",)
        else:
            ch = c
        if in_class:
            return ch, False
        return self._char(ch)

    def _class(self):
        neg = self._peek() == "^"
        if neg:
            self.i += 1
        mask, wide, multibyte = 0, False, []
        first = True
        while True:
            if self.i >= len(self.p):
                raise ValueError("missing ']' in regex")
            c = self.p[self.i]
            if c == "]" and not first:
                self.i += 1
                break
            first = False
            self.i += 1
            if c == "\\":
                v, w = self._escape(in_class=True)
                if isinstance(v, int):
                    mask |= v
                    wide |= w
                    continue
                c = v
            if self._peek() == "-" and self.i + 1 < len(self.p) and self.p[self.i + 1] != "]":
                self.i += 1
                hi = self.p[self.i]
                self.i += 1
                if hi == "\\":
                    hi, _ = self._escape(in_class=True)
                lo_o, hi_o = ord(c), ord(hi)
                if hi_o < 128:
                    mask |= _range(lo_o, hi_o)
                else:
                    if lo_o < 128:
                        mask |= _range(lo_o, 127)
                    wide = True                 # coarse: a non-ASCII range admits any multi-byte char
            elif ord(c) < 128:
                mask |= 1 << ord(c)
            else:
                multibyte.append(c)
        if neg:
            return ("alt", [("lit", ASCII & ~mask), ("utf8",)])
        alts = [("lit", mask)] if mask else []
        if wide:
            alts.append(("utf8",))
        alts += [self._char(c) for c in multibyte]
        if not alts:
            raise ValueError("empty character class")
        return alts[0] if len(alts) == 1 else ("alt", alts)


# ---- NFA ---------------------------------------------------------------------------------------
class _NFA:
    def __init__(self):
        self.eps: list[list[int]] = []
        self.trans: list[list[tuple[int, int]]] = []

    def new(self) -> int:
        self.eps.append([])
        self.trans.append([])
        return len(self.eps) - 1

    def build(self, node) -> tuple[int, int]:
        kind = node[0]
        if kind == "eps":
            s = self.new()
            return s, s
        if kind == "lit":
            s, e = self.new(), self.new()
            self.trans[s].append((node[1], e))
            return s, e
        if kind == "utf8":
            s, e = self.new(), self.new()
            c1, c2 = self.new(), self.new()
            cont = _range(0x80, 0xBF)
            self.trans[s].append((_range(0xC2, 0xDF), c1 if False else -1))     # placeholder, fixed below
            self.trans[s].pop()
            t1 = self.new()                       # needs 1 continuation byte
            self.trans[s].append((_range(0xC2, 0xDF), t1))
            self.trans[s].append((_range(0xE0, 0xEF), c1))    # needs 2
            self.trans[s].append((_range(0xF0, 0xF4), c2))    # needs 3
            self.trans[c2].append((cont, c1))
            self.trans[c1].append((cont, t1))
            self.trans[t1].append((cont, e))
            return s, e
        if kind == "cat":
            s, e = self.build(node[1][0])
            for sub in node[1][1:]:
                s2, e2 = self.build(sub)
                self.eps[e].append(s2)
                e = e2
            return s, e
        if kind == "alt":
            s, e = self.new(), self.new()
            for sub in node[1]:
                s2, e2 = self.build(sub)
                self.eps[s].append(s2)
                self.eps[e2].append(e)
            return s, e
        if kind == "rep":
            _, sub, lo, hi = node
            if hi is not None and hi > 1000 or lo > 1000:
                raise ValueError("regex repetition count too large")
            s = self.new()
            e = s
            for _ in range(lo):
                s2, e2 = self.build(sub)
                self.eps[e].append(s2)
                e = e2
            if hi is None:
                s2, e2 = self.build(sub)
                loop = self.new()
                self.eps[e].append(loop)
                self.eps[loop].append(s2)
                self.eps[e2].append(loop)
                e = loop
            else:
                end = self.new()
                self.eps[e].append(end)
                for _ in range(hi - lo):
                    s2, e2 = self.build(sub)
                    self.eps[e].append(s2)
                    self.eps[e2].append(end)
                    e = e2
                e = end
            return s, e
        raise AssertionError(kind)


class DFA:
    """trans[state, byte] -> state; ``dead`` is an absorbing non-accepting state."""

    def __init__(self, trans: np.ndarray, accepting: np.ndarray, start: int, dead: int):
        self.trans, self.accepting, self.start, self.dead = trans, accepting, start, dead

    @property
    def num_states(self) -> int:
        return self.trans.shape[0]

    def matches(self, data: bytes) -> bool:
        s = self.start
        for b in data:
            s = int(self.trans[s, b])
        return bool(self.accepting[s])


def compile_regex(pattern: str, max_states: int = 200_000, extra_symbols: int = 0) -> DFA:
    nfa = _NFA()
    start, end = nfa.build(_Parser(pattern).parse())
    n = len(nfa.eps)
    # byte equivalence classes over every mask in use
    masks = sorted({m for tr in nfa.trans for m, _ in tr})
    sig_to_class: dict[tuple, int] = {}
    nsym = 256 + extra_symbols
    byte_class = np.zeros(nsym, dtype=np.int32)
    reps: list[int] = []
    for b in range(nsym):
        sig = tuple((m >> b) & 1 for m in masks)
        if sig not in sig_to_class:
            sig_to_class[sig] = len(reps)
            reps.append(b)
        byte_class[b] = sig_to_class[sig]
    # epsilon closures
    closure_cache: dict[int, frozenset] = {}

    def closure(states) -> frozenset:
        out, stack = set(states), list(states)
        while stack:
            s = stack.pop()
            for t in nfa.eps[s]:
                if t not in out:
                    out.add(t)
                    stack.append(t)
        return frozenset(out)

    s0 = closure([start])
    ids = {s0: 0}
    order = [s0]
    rows: list[list[int]] = []
    i = 0
    while i < len(order):
        cur = order[i]
        i += 1
        row = []
        for rb in reps:
            bit = 1 << rb
            nxt = {t for s in cur for m, t in nfa.trans[s] if m & bit}
            if not nxt:
                row.append(-1)
                continue
            key = frozenset(nxt)
            cl = closure_cache.get(key)
            if cl is None:
                cl = closure_cache[key] = closure(nxt)
            j = ids.get(cl)
            if j is None:
                j = ids[cl] = len(order)
                order.append(cl)
                if j > max_states:
                    raise ValueError("guided-decoding automaton is too large")
            row.append(j)
        rows.append(row)
    ns = len(order)
    table = np.asarray(rows, dtype=np.int32).reshape(ns, len(reps))
    table[table < 0] = ns
    table = np.vstack([table, np.full((1, len(reps)), ns, dtype=np.int32)])
    trans = table[:, byte_class]                      # [ns + 1, 256 (+ extra symbols)]
    accepting = np.zeros(ns + 1, dtype=bool)
    for st, j in ids.items():
        if end in st:
            accepting[j] = True
    return DFA(np.ascontiguousarray(trans), accepting, 0, ns)
