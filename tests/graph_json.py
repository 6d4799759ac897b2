"""Test infrastructure: canonical JSON form of a graph recorded by rangedet_amd.mx (nodes in depth-first input order; op, explicit
name, attributes, input edges), used to compare the graph the REFERENCE's model code records (tests/golden/graph_veh_test.json,
made by tests/golden/make_ref_python_golden.py) with the one this package's mirror of that code records."""
import re

import numpy as np


def _plain(v):
    if isinstance(v, (tuple, list)):
        return [_plain(x) for x in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating, float)):
        return float(v)
    if isinstance(v, (bool, int, str)) or v is None:
        return v
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in sorted(v.items())}
    return repr(v)


def graph_to_json(sym):
    """List of nodes; node = {"op", "name" (None when auto-generated), "attrs", "inputs": [[node index, output index], ...]}."""
    index, nodes = {}, []

    def walk(s):
        if s.uid in index:
            return index[s.uid]
        ins = [[walk(i), int(i.index)] for i in s.inputs]
        auto = re.fullmatch(re.escape(s.op.lower()) + r"\d+", s.name) is not None
        index[s.uid] = len(nodes)
        attrs = s.attrs
        if s.op == "var":     # training-time decoration of a variable (initialiser object, lr / wd multipliers) is not graph structure
            attrs = {k: v for k, v in attrs.items() if k not in ("init", "lr_mult", "wd_mult", "param")}
        nodes.append({"op": s.op, "name": None if auto else s.name, "attrs": _plain(attrs), "inputs": ins})
        return index[s.uid]
    walk(sym)
    return nodes


def first_difference(a, b):
    """Human-readable location of the first node where two graph JSONs differ (None if equal)."""
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return "node %d: %r != %r" % (i, x, y)
    if len(a) != len(b):
        return "node count %d != %d" % (len(a), len(b))
    return None
