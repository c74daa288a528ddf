"""JSON schema -> regular expression (the subset guided decoding needs): type string / integer /
number / boolean / null / array / object, enum, const, anyOf / oneOf, local $ref, string
pattern / minLength / maxLength / format (date, time, date-time, uuid), numeric digit bounds,
array minItems / maxItems, required vs optional properties (kept in declaration order).
Unconstrained values (``{}`` or ``json_object``) use a JSON grammar of bounded nesting depth.
Same role as the schema front ends of the reference's backends
(vllm/v1/structured_output/backend_{xgrammar,outlines,guidance}.py)."""
from __future__ import annotations

import json
import re
from typing import Any

STRING_INNER = r'([^"\\\x00-\x1f]|\\["\\/bfnrt]|\\u[0-9a-fA-F]{4})'
STRING = f'"{STRING_INNER}*"'
INTEGER = r"(-)?(0|[1-9][0-9]*)"
NUMBER = rf"({INTEGER})(\.[0-9]+)?([eE][+-]?[0-9]+)?"
BOOLEAN = r"(true|false)"
NULL = r"null"
FORMATS = {
    "uuid": r'"[0-9a-fA-F]{8}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{12}"',
    "date": r'"[0-9]{4}-(0[1-9]|1[0-2])-(0[1-9]|[12][0-9]|3[01])"',
    "time": r'"([01][0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9](\.[0-9]+)?(Z|[+-][0-9]{2}:[0-9]{2})?"',
    "date-time": r'"[0-9]{4}-(0[1-9]|1[0-2])-(0[1-9]|[12][0-9]|3[01])T([01][0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]'
                 r'(\.[0-9]+)?(Z|[+-][0-9]{2}:[0-9]{2})?"',
}


def _lit(value: Any) -> str:
    return re.escape(json.dumps(value, ensure_ascii=False))


def any_json(depth: int, ws: str) -> str:
    scalar = f"({STRING}|{NUMBER}|{BOOLEAN}|{NULL})"
    value = scalar
    for _ in range(depth):
        arr = rf"\[{ws}({value}({ws},{ws}{value})*)?{ws}\]"
        obj = rf"\{{{ws}({STRING}{ws}:{ws}{value}({ws},{ws}{STRING}{ws}:{ws}{value})*)?{ws}\}}"
        value = f"({scalar}|{arr}|{obj})"
    return value


class SchemaCompiler:
    def __init__(self, any_whitespace: bool = True, max_depth: int = 3):
        self.ws = r"[ \n\t]{0,8}" if any_whitespace else ""
        self.sep_ws = r"[ ]?" if any_whitespace else ""
        self.max_depth = max_depth

    def compile(self, schema) -> str:
        if isinstance(schema, str):
            schema = json.loads(schema)
        self.root = schema
        self._ref_stack: list[str] = []
        return self._value(schema)

    def _resolve(self, ref: str) -> dict:
        if not ref.startswith("#"):
            raise ValueError(f"only local $ref is supported, got {ref!r}")
        node = self.root
        for part in ref.lstrip("#/").split("/"):
            if part:
                node = node[part.replace("~1", "/").replace("~0", "~")]
        return node

    def _value(self, s) -> str:
        ws = self.ws
        if s is True or s == {} or s is None:
            return any_json(self.max_depth, self.sep_ws)
        if "$ref" in s:
            ref = s["$ref"]
            if self._ref_stack.count(ref) >= 2:
                raise ValueError(f"recursive schema reference {ref!r} is not supported")
            self._ref_stack.append(ref)
            try:
                return self._value(self._resolve(ref))
            finally:
                self._ref_stack.pop()
        if "const" in s:
            return _lit(s["const"])
        if "enum" in s:
            return "(" + "|".join(_lit(v) for v in s["enum"]) + ")"
        for key in ("anyOf", "oneOf"):
            if key in s:
                return "(" + "|".join(self._value(x) for x in s[key]) + ")"
        if "allOf" in s:
            merged: dict = {}
            for x in s["allOf"]:
                x = self._resolve(x["$ref"]) if "$ref" in x else x
                for k, v in x.items():
                    if k == "properties":
                        merged.setdefault(k, {}).update(v)
                    elif k == "required":
                        merged[k] = list(merged.get(k, [])) + list(v)
                    else:
                        merged[k] = v
            return self._value(merged)
        t = s.get("type")
        if isinstance(t, list):
            return "(" + "|".join(self._value(dict(s, type=x)) for x in t) + ")"
        if t is None:
            if "properties" in s:
                t = "object"
            elif "items" in s or "prefixItems" in s:
                t = "array"
            else:
                return any_json(self.max_depth, self.sep_ws)
        if t == "string":
            if "pattern" in s:
                pat = s["pattern"]
                pat = pat[1:] if pat.startswith("^") else pat
                pat = pat[:-1] if pat.endswith("$") and not pat.endswith("\\$") else pat
                return f'"({pat})"'
            if s.get("format") in FORMATS:
                return FORMATS[s["format"]]
            lo, hi = s.get("minLength"), s.get("maxLength")
            if lo is not None or hi is not None:
                return f'"{STRING_INNER}{{{lo or 0},{"" if hi is None else hi}}}"'
            return STRING
        if t == "integer":
            return self._digits(s, INTEGER)
        if t == "number":
            return NUMBER
        if t == "boolean":
            return BOOLEAN
        if t == "null":
            return NULL
        if t == "array":
            return self._array(s)
        if t == "object":
            return self._object(s)
        raise ValueError(f"unsupported schema type {t!r}")

    @staticmethod
    def _digits(s, default: str) -> str:
        lo, hi = s.get("minDigits"), s.get("maxDigits")
        if lo is None and hi is None:
            return default
        lo = max(int(lo or 1), 1)
        return rf"(-)?(0|[1-9][0-9]{{{lo - 1},{'' if hi is None else int(hi) - 1}}})"

    def _array(self, s) -> str:
        ws = self.sep_ws
        if "prefixItems" in s:
            items = [self._value(x) for x in s["prefixItems"]]
            return r"\[" + ws + (ws + "," + ws).join(items) + ws + r"\]"
        item = self._value(s.get("items", {}))
        lo, hi = int(s.get("minItems", 0)), s.get("maxItems")
        if hi is not None and int(hi) == 0:
            return rf"\[{ws}\]"
        more = f"({ws},{ws}{item})"
        if lo == 0:
            rest = f"{{0,{int(hi) - 1}}}" if hi is not None else "*"
            return rf"\[{ws}({item}{more}{rest})?{ws}\]"
        rest = f"{{{lo - 1},{'' if hi is None else int(hi) - 1}}}"
        return rf"\[{ws}{item}{more}{rest}{ws}\]"

    def _object(self, s) -> str:
        ws = self.sep_ws
        props = s.get("properties") or {}
        if not props:
            ap = s.get("additionalProperties")
            if isinstance(ap, dict):
                v = self._value(ap)
                pair = f"{STRING}{ws}:{ws}{v}"
                return rf"\{{{ws}({pair}({ws},{ws}{pair})*)?{ws}\}}"
            return any_json(max(self.max_depth - 1, 1), ws) if ap is not False and not s.get("required") \
                else rf"\{{{ws}\}}"
        required = set(s.get("required", []))
        names = list(props)
        pairs = [f'{_lit(n)}{ws}:{ws}{self._value(props[n])}' for n in names]
        is_req = [n in required for n in names]
        # Properties keep their declaration order.  After the first emitted property every later one
        # carries its own leading comma; optional ones are wrapped in (...)?.
        if any(is_req):
            first = is_req.index(True)
            body = ""
            for i in range(first):
                body += f"({pairs[i]}{ws},{ws})?"
            body += pairs[first]
            for i in range(first + 1, len(names)):
                part = f"{ws},{ws}{pairs[i]}"
                body += part if is_req[i] else f"({part})?"
            return rf"\{{{ws}{body}{ws}\}}"
        # all optional: any non-empty ordered subset, or nothing
        alts = []
        for i in range(len(names)):
            tail = "".join(f"({ws},{ws}{pairs[j]})?" for j in range(i + 1, len(names)))
            alts.append(pairs[i] + tail)
        return rf"\{{{ws}(" + "|".join(alts) + rf")?{ws}\}}"


def schema_to_regex(schema, any_whitespace: bool = True) -> str:
    return SchemaCompiler(any_whitespace).compile(schema)
