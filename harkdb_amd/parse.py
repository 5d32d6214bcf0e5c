"""SQL statement -> planner IR.  Mirrors the reference's parse.py:9-91.

For the statements the reference understands the IR is identical
(SURVEY.md 8(b)):
    "select col1, col3 from game_1"
        -> {"table": <data>, "select": [0, 2]}                         (parse.py:58)
    "select col1, max(col3) from game_1 group by col1"
        -> {"select": [0, 2], "groupbys": [0, 3], "table": <data>, "g_col": 0}   (parse.py:90)
Extension clauses add keys ("where", "having", "orderby", "limit", "items",
"extended"); the reference ignores those clauses (parse.py only reads the
`select`, `from` and `groupby` keys).
"""
from .sqlfront import parse, AGGREGATES

# parse.py:81 -- the reference's opcode table, plus the extensions of include/hark.h
funcToFut = {"prod": 1, "sum": 2, "max": 3, "min": 4}
EXT_FUNCS = {"count": 5, "avg": 6}
_CMP_SQL = {"gt": ">", "gte": ">=", "lt": "<", "lte": "<=", "eq": "=", "neq": "!="}


def getIndex(elements, value):
    # parse.py:9-13
    for i, v in enumerate(elements):
        if v == value:
            return i
    return -1


def _col(columns, name, table_name):
    idx = getIndex(columns, name)
    if idx < 0 and isinstance(name, str) and name.startswith(table_name + "."):   # `t.col` in a one-table statement
        idx = getIndex(columns, name[len(table_name) + 1:])
    if idx < 0:
        raise Exception(f"{name} is not in the schema of table {table_name}")     # parse.py:54
    return idx


_ARITH = ("add", "sub", "mul", "div")


def expr_node(arg, col_of):
    """The argument of an aggregate in moz shape -> ("col", j) | ("num", v) | (op, left, right)."""
    if isinstance(arg, str):
        return ("col", col_of(arg))
    if isinstance(arg, (int, float)):
        return ("num", arg)
    (op, (l, r)), = arg.items()
    if op not in _ARITH:
        raise Exception(f"{op} is not an arithmetic operator (+ - * /)")
    return (op, expr_node(l, col_of), expr_node(r, col_of))


def expr_text(node, columns):
    if node[0] == "col":
        return columns[node[1]]
    if node[0] == "num":
        return repr(node[1])
    return "(" + expr_text(node[1], columns) + {"add": " + ", "sub": " - ", "mul": " * ", "div": " / "}[node[0]] + expr_text(node[2], columns) + ")"


def _is_simple(cond):
    """`column <op> number`: what the reference-shaped AND-lists are made of."""
    (op, args), = cond.items()
    return op in _CMP_SQL and isinstance(args[0], str) and isinstance(args[1], (int, float))


def _unqualify(js_obj, table_name, columns):
    """`t.col` -> `col` in the select list, GROUP BY, HAVING and ORDER BY of a one-table statement over t (WHERE resolves its
    names through _col, which knows the prefix too)."""
    pre = table_name + "."

    def name(x):
        return x[len(pre):] if isinstance(x, str) and x.startswith(pre) and x not in columns and x[len(pre):] in columns else x

    def value(v):
        if isinstance(v, dict) and len(v) == 1:
            (f, a), = v.items()
            return {f: name(a)}
        return name(v)

    def items(x):
        for it in (x if isinstance(x, list) else [x]):
            if isinstance(it, dict) and "value" in it:
                it["value"] = value(it["value"])
    for key in ("select", "groupby", "orderby"):
        if key in js_obj:
            items(js_obj[key])
    if "having" in js_obj:
        js_obj["having"] = _rename_terms(js_obj["having"], value)


def _rename_terms(cond, f):
    (op, args), = cond.items()
    if op in ("and", "or"):
        return {op: [_rename_terms(c, f) for c in args]}
    if op == "not":
        return {op: _rename_terms(args, f)}
    return {op: [f(args[0])] + list(args[1:])}


def where_conjuncts(tree, col_of):
    """A WHERE tree (moz shape) -> the IR's conjunct list: (column, cmp, number) for plain comparisons -- exactly what the AND-lists
    of earlier rounds gave -- and (None, "tree", node) for everything else (OR / NOT / IN / parentheses / two columns), node =
    ("and" | "or", [nodes]) | ("not", node) | ("cmp", col, cmp, number) | ("cmpcol", col, cmp, col2) | ("in", col, [numbers]);
    the executor evaluates a node into a survivor bitmask on the device (Engine.predicate_tree_mask) and hands it to the
    kernels as one more conjunct.  col_of(name) -> column index."""
    def node(cond):
        (op, args), = cond.items()
        if op in ("and", "or"):
            return (op, [node(c) for c in args])
        if op == "not":
            return ("not", node(args))
        if op in ("in", "nin"):
            lhs, vals = args
            vals = vals if isinstance(vals, list) else [vals]
            if not isinstance(lhs, str) or not all(isinstance(v, (int, float)) for v in vals):
                raise Exception("IN takes a column and a list of numbers")
            inner = ("in", col_of(lhs), list(vals))
            return ("not", inner) if op == "nin" else inner
        if op not in _CMP_SQL:
            raise Exception(f"{op} is not a supported comparison")
        lhs, rhs = args
        if isinstance(lhs, str) and isinstance(rhs, (int, float)):
            return ("cmp", col_of(lhs), _CMP_SQL[op], rhs)
        if isinstance(lhs, str) and isinstance(rhs, str):
            return ("cmpcol", col_of(lhs), _CMP_SQL[op], col_of(rhs))
        raise Exception("WHERE compares a column with a number or with another column")

    out = []
    for cond in _conditions(tree):
        if _is_simple(cond):
            (op, (lhs, rhs)), = cond.items()
            out.append((col_of(lhs), _CMP_SQL[op], rhs))
        else:
            out.append((None, "tree", node(cond)))
    return out


def _names_of(cond, acc):
    """Column names a condition mentions."""
    (op, args), = cond.items()
    if op in ("and", "or"):
        for c in args:
            _names_of(c, acc)
    elif op == "not":
        _names_of(args, acc)
    else:
        for x in (args[:1] if op in ("in", "nin") else args):
            if isinstance(x, str):
                acc.append(x)
    return acc


def _rename(cond, f):
    """The condition with every column name x replaced by f(x)."""
    (op, args), = cond.items()
    if op in ("and", "or"):
        return {op: [_rename(c, f) for c in args]}
    if op == "not":
        return {op: _rename(args, f)}
    if op in ("in", "nin"):
        return {op: [f(args[0]), args[1]]}
    return {op: [f(x) if isinstance(x, str) else x for x in args]}


def _conditions(tree):
    if tree is None:
        return []
    return tree["and"] if "and" in tree else [tree]


def sql_parse(tables, sql_statement):
    """Parses an SQL statement (parse.py:16)."""
    return sql_parse_tree(tables, parse(sql_statement))                 # parse.py:27


def sql_parse_tree(tables, js_obj):
    """The planner proper: parse tree (moz_sql_parser's JSON shape) -> IR."""
    if "select_distinct" in js_obj:
        # SELECT DISTINCT a, b  ==  SELECT a, b ... GROUP BY a, b  (extension; rows come out in ascending key order)
        sel = js_obj.pop("select_distinct")
        sel = [sel] if isinstance(sel, (dict, str)) else sel
        if "groupby" in js_obj or any(s == "*" or not isinstance(s["value"], str) for s in sel):
            raise Exception("SELECT DISTINCT takes plain columns and no GROUP BY")
        js_obj["select"] = sel
        keys = list(dict.fromkeys(s["value"] for s in sel))
        js_obj["groupby"] = {"value": keys[0]} if len(keys) == 1 else [{"value": k} for k in keys]
        distinct = True
    else:
        distinct = False
    if isinstance(js_obj["from"], list):                                # two-table FROM (extension)
        return _join_parse(tables, js_obj)
    table_name = js_obj["from"]
    if table_name in tables:                                            # parse.py:30-33
        table = tables[table_name]
    else:
        raise Exception(f"{table_name} is not in tables")
    columns = table.get_schema()                                        # parse.py:40
    _unqualify(js_obj, table_name, columns)
    select_pairs = js_obj["select"]
    if isinstance(select_pairs, (dict, str)):                           # one item / "*": the reference
        select_pairs = [select_pairs]                                   # breaks here (TypeError, parse.py:50)
    ir = {"table": table.get_data(), "table_name": table_name, "extended": False}

    # ---- WHERE (extension; the reference ignores the key) ------------------
    where = where_conjuncts(js_obj.get("where"), lambda name: _col(columns, name, table_name))
    if where:
        ir["where"] = where
        ir["extended"] = True

    if "groupby" not in js_obj.keys():                                  # parse.py:42
        fut_cols_selects, items = [], []
        for pair in select_pairs:
            if pair == "*":
                fut_cols_selects += list(range(len(columns)))
                items += [("col", i) for i in range(len(columns))]
            elif "value" in pair:                                       # parse.py:50
                if not isinstance(pair["value"], str):
                    raise Exception("aggregates need a GROUP BY clause")
                idx = _col(columns, pair["value"], table_name)
                fut_cols_selects += [idx]
                items.append(("col", idx))
        ir["select"] = fut_cols_selects                                 # parse.py:58
        ir["items"] = items
    else:                                                               # parse.py:60
        fut_cols_selects, typ_cols_selects, items = [], [], []

        def derived_col(arg):
            """The column number of an aggregate's arithmetic argument: behind the table's own columns, one per distinct expression
            (the executor computes them on the device before the statement runs, Engine.column_expr)."""
            node = expr_node(arg, lambda name: _col(columns, name, table_name))
            der = ir.setdefault("derived", [])
            if node not in der:
                der.append(node)
            return len(columns) + der.index(node)
        gb = js_obj["groupby"]
        g_names = [g["value"] for g in gb] if isinstance(gb, list) else [gb["value"]]   # several keys: extension
        g_col_name = g_names[0]                                         # parse.py:66
        g_cols = []
        for name in g_names:
            idx = getIndex(columns, name)
            if idx < 0:
                raise Exception(f"{name} is not in the schema of table {table_name}")
            if idx in g_cols:
                raise Exception(f"{name} is grouped on twice")
            g_cols.append(idx)
        g_col = g_cols[0]
        if len(g_cols) > 1:
            ir["g_cols"] = g_cols
            ir["extended"] = True
        for dic in select_pairs:                                        # parse.py:72
            if dic == "*":
                raise Exception("* is not allowed with GROUP BY")
            if isinstance(dic["value"], str) and dic["value"] in g_names:   # parse.py:73-75
                kc = g_cols[g_names.index(dic["value"])]
                fut_cols_selects += [kc]
                typ_cols_selects += [0]
                items.append(("key", kc))
            elif isinstance(dic["value"], str):                         # parse.py:76-78
                bad_col_name = dic["value"]
                raise Exception(f"{bad_col_name} is not an aggregation function or the columns thats grouped on")
            else:
                (agg_func, agg_col_name), = dic["value"].items()
                if isinstance(agg_col_name, dict) and "distinct" in agg_col_name:      # count(distinct x) (extension)
                    if agg_func != "count":
                        raise Exception("DISTINCT is supported inside count() only")
                    items.append(("count_distinct", _col(columns, agg_col_name["distinct"], table_name)))
                    ir["extended"] = True
                    continue
                if isinstance(agg_col_name, (dict, int, float)):        # arithmetic inside the aggregate (extension): a derived column
                    if agg_func not in funcToFut and agg_func not in EXT_FUNCS:
                        raise Exception(f"{agg_func} is not a supported aggregation function {AGGREGATES}")
                    items.append((agg_func, derived_col(agg_col_name)))
                    ir["extended"] = True
                    continue
                if agg_func in funcToFut:                               # parse.py:82-89
                    agg_col = _col(columns, agg_col_name, table_name)
                    fut_cols_selects += [agg_col]
                    typ_cols_selects += [funcToFut[agg_func]]
                    items.append((agg_func, agg_col))
                elif agg_func in EXT_FUNCS:                             # extension: COUNT / AVG
                    agg_col = None if agg_col_name == "*" else _col(columns, agg_col_name, table_name)
                    items.append((agg_func, agg_col))
                    ir["extended"] = True
                else:                                                   # the reference drops these silently
                    raise Exception(f"{agg_func} is not a supported aggregation function {AGGREGATES}")
        ir["select"] = fut_cols_selects                                 # parse.py:90
        ir["groupbys"] = typ_cols_selects
        ir["g_col"] = g_col
        ir["items"] = items
        if distinct:
            ir["extended"] = True

    # ---- HAVING / ORDER BY / LIMIT (extensions) ------------------------------
    aliases = {p["name"]: p["value"] for p in select_pairs if isinstance(p, dict) and "name" in p}      # `sum(col2) as s ... order by s`

    def spec_of(term):
        """A key / aggregate reference in HAVING or ORDER BY -> the matching item."""
        if isinstance(term, str) and term in aliases and getIndex(columns, term) < 0:
            term = aliases[term]
        if isinstance(term, str):
            idx = _col(columns, term, table_name)
            return ("key", idx) if "groupby" in js_obj and idx in ir.get("g_cols", [ir["g_col"]]) else ("col", idx)
        (f, c), = term.items()
        if f not in funcToFut and f not in EXT_FUNCS:
            raise Exception(f"{f} is not a supported aggregation function {AGGREGATES}")
        if isinstance(c, dict) and "distinct" in c:
            raise Exception("count(distinct ...) is not supported in HAVING / ORDER BY")
        if isinstance(c, (dict, int, float)):                           # the same expression as in the select list (or a new one)
            if "groupby" not in js_obj:
                raise Exception("aggregates need a GROUP BY clause")
            node = expr_node(c, lambda name: _col(columns, name, table_name))
            der = ir.setdefault("derived", [])
            if node not in der:
                der.append(node)
            return (f, len(columns) + der.index(node))
        return (f, None if c == "*" else _col(columns, c, table_name))

    having = []
    for cond in _conditions(js_obj.get("having")):
        (op, args), = cond.items()
        if op not in _CMP_SQL or not isinstance(args[1], (int, float)):
            raise Exception("HAVING supports `aggregate <op> number` comparisons (joined by AND)")
        lhs, rhs = args
        having.append((spec_of(lhs), _CMP_SQL[op], rhs))
    if having:
        if "groupby" not in js_obj:
            raise Exception("HAVING needs a GROUP BY clause")
        ir["having"] = having
        ir["extended"] = True
    if "orderby" in js_obj:
        obs = js_obj["orderby"] if isinstance(js_obj["orderby"], list) else [js_obj["orderby"]]
        ir["orderby"] = (spec_of(obs[0]["value"]), obs[0].get("sort") == "desc")
        if len(obs) > 1:                                                # several sort keys (extension of the extension)
            ir["orderby_all"] = [(spec_of(o["value"]), o.get("sort") == "desc") for o in obs]
        ir["extended"] = True
    if "limit" in js_obj:
        ir["limit"] = int(js_obj["limit"])
        ir["extended"] = True
    return ir


def _qualified(name, names, tables):
    """`table.column` -> (side, column index); a bare column must be unambiguous."""
    if "." in name:
        t, c = name.split(".", 1)
        if t not in names:
            raise Exception(f"{t} is not in tables")
        side = names.index(t)
        return side, _col(tables[t].get_schema(), c, t)
    hits = [(i, getIndex(tables[t].get_schema(), name)) for i, t in enumerate(names)]
    hits = [(i, j) for i, j in hits if j >= 0]
    if len(hits) != 1:
        raise Exception(f"{name} is {'ambiguous' if hits else 'not in the schema of table ' + ' or '.join(names)}")
    return hits[0]


def _join_parse(tables, js_obj):
    """select a.x, b.y from a join b on a.k = b.k  ->  the arguments of `entry join`
    (futhark/join.fut:52-54): col1, col2, cols1, cols2 (+ the select order)."""
    left, j = js_obj["from"]
    right = j["inner join"]
    for t in (left, right):
        if t not in tables:
            raise Exception(f"{t} is not in tables")                    # parse.py:33
    names = [left, right]
    (s1, c1), (s2, c2) = (_qualified(x, names, tables) for x in j["on"]["eq"])
    if s1 == s2:
        raise Exception("JOIN ... ON must compare a column of each table")
    if s1 == 1:
        c1, c2 = c2, c1
    sel = js_obj["select"]
    sel = [sel] if isinstance(sel, (dict, str)) else sel
    if any(cl in js_obj for cl in ("where", "groupby", "having", "orderby")) or any(it != "*" and not isinstance(it["value"], str) for it in sel):
        return _join_with_clauses(tables, js_obj, names, c1, c2, sel)
    order = []                                                          # (side, column) in select-list order
    for item in sel:
        if item == "*":
            order += [(0, i) for i in range(len(tables[left].get_schema()))] + [(1, i) for i in range(len(tables[right].get_schema()))]
        elif isinstance(item["value"], str):
            order.append(_qualified(item["value"], names, tables))
        else:
            raise Exception("aggregates are not supported together with JOIN")
    ir = {"join": True, "tables": names, "col1": c1, "col2": c2, "extended": True,
          "cols1": [c for s, c in order if s == 0], "cols2": [c for s, c in order if s == 1], "order": order}
    if "limit" in js_obj:
        ir["limit"] = int(js_obj["limit"])
    return ir


JOIN_RESULT = "__join__"          # the name under which a join's result is queried by the clauses around it


def _join_with_clauses(tables, js_obj, names, c1, c2, sel):
    """JOIN with WHERE / GROUP BY / HAVING / ORDER BY / aggregates (the reference has neither; SURVEY.md 8(f) 3 asks for the
    two-table FROM only).  Planned as three steps the executor runs on the device:
      1. WHERE is an AND-list of `column <op> number` comparisons, each on ONE table: every conjunct is pushed below the
         join (a compaction keeps row order, so the join's (key, left row, right row) order is the same as filtering after);
      2. the join delivers every column the other clauses mention, as a table whose schema holds the QUALIFIED names
         (`t.k`, `b.y`);
      3. "post": the statement's remaining clauses as a single-table parse tree over that table (column references
         rewritten to their qualified names), planned by sql_parse_tree like any other statement."""
    def qname(side, col):
        return f"{names[side]}.{tables[names[side]].get_schema()[col]}"

    used = []                                                           # (side, column) the join has to deliver, in first-use order

    def ref(name):
        sc = _qualified(name, names, tables)
        if sc not in used:
            used.append(sc)
        return qname(*sc)

    def term(t):                                                        # column | {agg: column | "*"} | number
        if isinstance(t, str):
            return t if t == "*" else ref(t)
        if isinstance(t, dict):
            (f, a), = t.items()
            return {f: a if a == "*" else ref(a)}
        return t

    where, mixed = [[], []], []
    for cond in _conditions(js_obj.get("where")):
        sides = {_qualified(x, names, tables)[0] for x in _names_of(cond, [])}
        if len(sides) == 1:                                             # a conjunct on ONE table (whatever its shape): below the join
            side = sides.pop()
            where[side] += where_conjuncts(cond, lambda name: _qualified(name, names, tables)[1])
        elif len(sides) == 2:                                           # it mentions both tables: a WHERE of the statement over the join's result
            mixed.append(_rename(cond, ref))
        else:
            raise Exception("a WHERE condition must mention a column")
    post = {"from": JOIN_RESULT}
    if mixed:
        post["where"] = mixed[0] if len(mixed) == 1 else {"and": mixed}
    items = []
    for it in sel:
        if it == "*":
            items += [{"value": ref(f"{names[sd]}.{c}")} for sd in (0, 1) for c in tables[names[sd]].get_schema()]
        else:
            items.append({"value": term(it["value"])})
    post["select"] = items[0] if len(items) == 1 else items
    if "groupby" in js_obj:
        gb = js_obj["groupby"]
        gb = [{"value": ref(g["value"])} for g in (gb if isinstance(gb, list) else [gb])]
        post["groupby"] = gb[0] if len(gb) == 1 else gb
    if "having" in js_obj:
        conds = [{op: [term(l), r]} for cond in _conditions(js_obj["having"]) for (op, (l, r)) in [next(iter(cond.items()))]]
        post["having"] = conds[0] if len(conds) == 1 else {"and": conds}
    if "orderby" in js_obj:
        obs = js_obj["orderby"] if isinstance(js_obj["orderby"], list) else [js_obj["orderby"]]
        obs = [dict(o, value=term(o["value"])) for o in obs]
        post["orderby"] = obs[0] if len(obs) == 1 else obs
    if "limit" in js_obj:
        post["limit"] = js_obj["limit"]
    return {"join": True, "tables": names, "col1": c1, "col2": c2, "extended": True,
            "cols1": [c for s_, c in used if s_ == 0], "cols2": [c for s_, c in used if s_ == 1],
            "where1": where[0], "where2": where[1], "post": post,
            "post_schema": [qname(s_, c) for s_, c in used if s_ == 0] + [qname(s_, c) for s_, c in used if s_ == 1]}
