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
    if idx < 0:
        raise Exception(f"{name} is not in the schema of table {table_name}")     # parse.py:54
    return idx


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
    select_pairs = js_obj["select"]
    if isinstance(select_pairs, (dict, str)):                           # one item / "*": the reference
        select_pairs = [select_pairs]                                   # breaks here (TypeError, parse.py:50)
    ir = {"table": table.get_data(), "table_name": table_name, "extended": False}

    # ---- WHERE (extension; the reference ignores the key) ------------------
    where = []
    for cond in _conditions(js_obj.get("where")):
        (op, (lhs, rhs)), = cond.items()
        if not isinstance(lhs, str) or not isinstance(rhs, (int, float)):
            raise Exception("WHERE supports `column <op> number` comparisons (joined by AND)")
        where.append((_col(columns, lhs, table_name), _CMP_SQL[op], rhs))
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
    def spec_of(term):
        """A key / aggregate reference in HAVING or ORDER BY -> the matching item."""
        if isinstance(term, str):
            idx = _col(columns, term, table_name)
            return ("key", idx) if "groupby" in js_obj and idx in ir.get("g_cols", [ir["g_col"]]) else ("col", idx)
        (f, c), = term.items()
        if f not in funcToFut and f not in EXT_FUNCS:
            raise Exception(f"{f} is not a supported aggregation function {AGGREGATES}")
        return (f, None if c == "*" else _col(columns, c, table_name))

    having = []
    for cond in _conditions(js_obj.get("having")):
        (op, (lhs, rhs)), = cond.items()
        if not isinstance(rhs, (int, float)):
            raise Exception("HAVING supports `aggregate <op> number` comparisons")
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

    where = [[], []]
    for cond in _conditions(js_obj.get("where")):
        (op, (lhs, rhs)), = cond.items()
        if not isinstance(lhs, str) or not isinstance(rhs, (int, float)):
            raise Exception("WHERE supports `column <op> number` comparisons (joined by AND)")
        side, col = _qualified(lhs, names, tables)
        where[side].append((col, _CMP_SQL[op], rhs))
    post = {"from": JOIN_RESULT}
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
