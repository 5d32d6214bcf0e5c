"""A small SQL front end that produces the parse tree `parse.py` consumes.

The reference calls `moz_sql_parser.parse` (parse.py:6, :27), an unpinned
third-party package that is not available here.  This module produces the same
JSON shape for the statements HarkDB understands (shape taken from parse.py's
own indexing: parse.py:29, :46-51, :66, :72-84):

    {"select": [{"value": "col1"}, {"value": {"max": "col3"}}],   # one item -> a dict, "*" -> "*"
     "from": "game_1",
     "where": {"gt": ["col2", 4]},                 # trees: {"or": [..]}, {"and": [..]}, {"not": x}, {"in": ["col1", [1, 2]]}, {"nin": ..}
     "groupby": {"value": "col1"},
     "having": {"gte": [{"sum": "col3"}, 10]},
     "orderby": {"value": "col1", "sort": "desc"},
     "limit": 10}
    two tables:  "from": ["a", {"inner join": "b", "on": {"eq": ["a.k", "b.k"]}}]
"""
import re

_TOKEN = re.compile(r"""\s*(?:
    (?P<num>[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?)
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*(?:\.[A-Za-z_][A-Za-z_0-9]*)?)
  | (?P<qid>"[^"]+"|`[^`]+`)
  | (?P<op><=|>=|<>|!=|==|=|<|>)
  | (?P<punct>[(),*])
  | (?P<arith>[-+/])
)""", re.X)
_ARITH = {"+": "add", "-": "sub", "*": "mul", "/": "div"}

_CMP = {">": "gt", ">=": "gte", "<": "lt", "<=": "lte", "=": "eq", "==": "eq", "!=": "neq", "<>": "neq"}
_FLIP = {"gt": "lt", "gte": "lte", "lt": "gt", "lte": "gte", "eq": "eq", "neq": "neq"}
AGGREGATES = ("prod", "sum", "max", "min", "count", "avg")
_KEYWORDS = {"select", "distinct", "from", "where", "group", "by", "having", "order", "limit", "asc", "desc", "as", "and", "or", "not", "in", "join", "inner", "on", "between"}


class SqlSyntaxError(Exception):
    pass


def _tokenize(text):
    pos, out = 0, []
    text = text.strip().rstrip(";")
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m or m.end() == pos:
            raise SqlSyntaxError(f"cannot tokenize SQL at: {text[pos:pos + 20]!r}")
        pos = m.end()
        if m.group("num") is not None:
            s = m.group("num")
            if s[0] in "+-" and out and (out[-1][0] in ("id", "num") or out[-1] == ("punct", ")")):
                out.append(("arith", s[0]))                  # `a -1`, `(a) +2`: the sign is the operator of an expression
                s = s[1:]
            out.append(("num", float(s) if any(c in s for c in ".eE") else int(s)))
        elif m.group("arith") is not None:
            out.append(("arith", m.group("arith")))
        elif m.group("id") is not None:
            w = m.group("id")
            out.append(("kw", w.lower()) if w.lower() in _KEYWORDS else ("id", w))
        elif m.group("qid") is not None:
            out.append(("id", m.group("qid")[1:-1]))
        elif m.group("op") is not None:
            out.append(("op", m.group("op")))
        else:
            out.append(("punct", m.group("punct")))
    return out


class _Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, kind=None, val=None):
        if self.i >= len(self.t):
            return False
        k, v = self.t[self.i]
        return (kind is None or k == kind) and (val is None or v == val)

    def take(self, kind=None, val=None):
        if not self.peek(kind, val):
            got = self.t[self.i] if self.i < len(self.t) else "end of statement"
            raise SqlSyntaxError(f"expected {val or kind}, got {got}")
        tok = self.t[self.i]
        self.i += 1
        return tok[1]

    def term(self):
        """column | agg(column) | agg(*) | number  ->  moz-style value."""
        if self.peek("num"):
            return self.take("num")
        if self.peek("punct", "*"):
            self.take()
            return "*"
        name = self.take("id")
        if self.peek("punct", "("):
            self.take()
            if self.peek("kw", "distinct"):                  # count(distinct x): moz's {"count": {"distinct": "x"}}
                self.take()
                arg = {"distinct": self.take("id")}
            elif self.peek("punct", "*"):
                self.take()
                arg = "*"
            else:
                arg = self.expression()                      # a column, or arithmetic over columns and numbers: sum(a + 2 * b)
            self.take("punct", ")")
            return {name.lower(): arg}
        return name

    # arithmetic inside an aggregate: + - * / over columns and numbers, parentheses; moz's {"add": [a, b]} ... shapes
    def atom(self):
        if self.peek("punct", "("):
            self.take()
            e = self.expression()
            self.take("punct", ")")
            return e
        if self.peek("arith", "-") or self.peek("arith", "+"):
            sign = self.take()
            v = self.atom()
            return v if sign == "+" else (-v if isinstance(v, (int, float)) else {"sub": [0, v]})
        if self.peek("num"):
            return self.take("num")
        return self.take("id")

    def product(self):
        e = self.atom()
        while self.peek("punct", "*") or self.peek("arith", "/"):
            e = {_ARITH[self.take()]: [e, self.atom()]}
        return e

    def expression(self):
        e = self.product()
        while self.peek("arith", "+") or self.peek("arith", "-"):
            e = {_ARITH[self.take()]: [e, self.product()]}
        return e

    def comparison(self):
        lhs = self.term()
        negated = self.peek("kw", "not")                     # x NOT BETWEEN ..., x NOT IN (...)
        if negated:
            self.take()
            if not (self.peek("kw", "between") or self.peek("kw", "in")):
                raise SqlSyntaxError("NOT must be followed by BETWEEN or IN here")
        if self.peek("kw", "between"):                       # x BETWEEN a AND b  ==  x >= a AND x <= b
            self.take()
            lo = self.term()
            self.take("kw", "and")
            hi = self.term()
            both = {"and": [{"gte": [lhs, lo]}, {"lte": [lhs, hi]}]}
            return {"not": both} if negated else both
        if self.peek("kw", "in"):                            # x IN (a, b, ...): moz's {"in": [x, [a, b, ...]]}
            self.take()
            self.take("punct", "(")
            vals = [self.take("num")]
            while self.peek("punct", ","):
                self.take()
                vals.append(self.take("num"))
            self.take("punct", ")")
            return {"nin" if negated else "in": [lhs, vals]}
        op = _CMP[self.take("op")]
        rhs = self.term()
        if isinstance(lhs, (int, float)) and not isinstance(rhs, (int, float)):
            lhs, rhs, op = rhs, lhs, _FLIP[op]
        return {op: [lhs, rhs]}

    def primary(self):
        if self.peek("punct", "("):                          # a term never starts with a parenthesis: this one groups a condition
            self.take()
            c = self.condition()
            self.take("punct", ")")
            return c
        return self.comparison()

    def negation(self):
        if self.peek("kw", "not"):
            self.take()
            return {"not": self.negation()}
        return self.primary()

    def conjunction(self):
        terms = []
        while True:
            c = self.negation()
            terms += c["and"] if "and" in c else [c]         # BETWEEN contributes two comparisons, (a AND b) AND c is one list
            if not self.peek("kw", "and"):
                break
            self.take()
        return terms[0] if len(terms) == 1 else {"and": terms}

    def condition(self):
        """OR of ANDs of (NOT) comparisons / parenthesised conditions -- the precedence of SQL."""
        terms = []
        while True:
            c = self.conjunction()
            terms += c["or"] if "or" in c else [c]
            if not self.peek("kw", "or"):
                break
            self.take()
        return terms[0] if len(terms) == 1 else {"or": terms}

    def statement(self):
        self.take("kw", "select")
        distinct = self.peek("kw", "distinct")
        if distinct:
            self.take()
        items = []
        while True:
            v = self.term()
            item = v if v == "*" else {"value": v}
            if self.peek("kw", "as"):
                self.take()
                item["name"] = self.take("id")
            items.append(item)
            if self.peek("punct", ","):
                self.take()
                continue
            break
        tree = {"select_distinct" if distinct else "select": items[0] if len(items) == 1 else items}
        self.take("kw", "from")
        tree["from"] = self.take("id")
        if self.peek("kw", "inner") or self.peek("kw", "join"):
            # moz-style: "from": [left, {"inner join": right, "on": {"eq": [l, r]}}]
            if self.peek("kw", "inner"):
                self.take()
            self.take("kw", "join")
            right = self.take("id")
            self.take("kw", "on")
            lhs = self.take("id")
            if self.take("op") not in ("=", "=="):
                raise SqlSyntaxError("JOIN ... ON supports equality only")
            rhs = self.take("id")
            tree["from"] = [tree["from"], {"inner join": right, "on": {"eq": [lhs, rhs]}}]
        if self.peek("kw", "where"):
            self.take()
            tree["where"] = self.condition()
        if self.peek("kw", "group"):
            self.take()
            self.take("kw", "by")
            keys = [{"value": self.take("id")}]
            while self.peek("punct", ","):                    # several keys -> a list (moz shape), one key -> a dict
                self.take()
                keys.append({"value": self.take("id")})
            tree["groupby"] = keys[0] if len(keys) == 1 else keys
        if self.peek("kw", "having"):
            self.take()
            tree["having"] = self.condition()
        if self.peek("kw", "order"):
            self.take()
            self.take("kw", "by")
            obs = []
            while True:
                ob = {"value": self.term()}
                if self.peek("kw", "desc"):
                    self.take()
                    ob["sort"] = "desc"
                elif self.peek("kw", "asc"):
                    self.take()
                obs.append(ob)
                if not self.peek("punct", ","):
                    break
                self.take()
            tree["orderby"] = obs[0] if len(obs) == 1 else obs      # several sort keys -> a list (moz shape)
        if self.peek("kw", "limit"):
            self.take()
            tree["limit"] = self.take("num")
        if self.i != len(self.t):
            raise SqlSyntaxError(f"unexpected trailing tokens: {self.t[self.i:]}")
        return tree


def parse(sql_statement):
    """Stand-in for `moz_sql_parser.parse` (parse.py:27)."""
    return _Parser(_tokenize(sql_statement)).statement()
