#!/usr/bin/env python
"""Generate metaworld_amd/csrc/mw_policies_gen.hpp from metaworld_amd/policies.py.

The 50 batched numpy policies are the single source of truth (they are pinned bit-exactly against the reference's scripted
policies by tests/test_batched_policies.py).  This script runs each of them once on a *symbolic* observation -- a (1, 39)
object array whose elements record the float64 arithmetic applied to them -- and prints the recorded expression DAG as one
straight-line C++ function per task: same operations, same order, same constants, so the device policy reproduces the numpy
policy bit for bit (tests/test_device_policies.py).  Usage: python tools/gen_device_policies.py [--check]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import policies as P, tasks as T  # noqa: E402

OUT = os.path.join(ROOT, "metaworld_amd", "csrc", "mw_policies_gen.hpp")


class Sym:
    """one recorded scalar: kind 'd' (double), 'b' (bool) or 'f' (float)"""
    table, order = {}, []

    def __new__(cls, op, *args, kind="d"):
        key = (op,) + tuple(a.key if isinstance(a, Sym) else ("c", repr(float(a)) if not isinstance(a, (bool, np.bool_)) else bool(a)) for a in args)
        if key in cls.table:
            return cls.table[key]
        s = object.__new__(cls)
        s.op, s.args, s.kind, s.key, s.name = op, args, kind, key, None
        cls.table[key] = s
        cls.order.append(s)
        return s

    @classmethod
    def reset(cls):
        cls.table, cls.order = {}, []

    # arithmetic (float64, one IEEE operation per node)
    def __add__(s, o): return Sym("+", s, o)
    def __radd__(s, o): return Sym("+", o, s)
    def __sub__(s, o): return Sym("-", s, o)
    def __rsub__(s, o): return Sym("-", o, s)
    def __mul__(s, o): return Sym("*", s, o)
    def __rmul__(s, o): return Sym("*", o, s)
    def __neg__(s): return Sym("neg", s)
    def __abs__(s): return Sym("fabs", s)
    def sqrt(s): return Sym("sqrt", s)
    # comparisons / logic
    def __gt__(s, o): return Sym(">", s, o, kind="b")
    def __lt__(s, o): return Sym("<", s, o, kind="b")
    def __ge__(s, o): return Sym(">=", s, o, kind="b")
    def __le__(s, o): return Sym("<=", s, o, kind="b")
    def __and__(s, o): return Sym("&&", s, o, kind="b")
    def __rand__(s, o): return Sym("&&", o, s, kind="b")
    def __or__(s, o): return Sym("||", s, o, kind="b")
    def __ror__(s, o): return Sym("||", o, s, kind="b")
    def __invert__(s): return Sym("!", s, kind="b")
    def __bool__(s): raise TypeError("symbolic value used as a Python bool")
    __hash__ = object.__hash__


def lit(v):
    if isinstance(v, Sym):
        return v.name
    if isinstance(v, (bool, np.bool_)):
        return "true" if v else "false"
    r = repr(float(v))
    return r if any(c in r for c in ".einf") else r + ".0"


class SymArr(np.ndarray):
    """object ndarray whose comparisons stay symbolic (numpy's default object comparison loops convert to bool)"""
    def __gt__(s, o): return np.greater(s, o, dtype=object)
    def __lt__(s, o): return np.less(s, o, dtype=object)
    def __ge__(s, o): return np.greater_equal(s, o, dtype=object)
    def __le__(s, o): return np.less_equal(s, o, dtype=object)


def is_sym(a):
    return isinstance(a, np.ndarray) and a.dtype == object


def obj(a):
    return (a if is_sym(a) else np.asarray(a, dtype=object)).view(SymArr)


def sym_where(c, x, y):
    c, x, y = np.broadcast_arrays(obj(c), obj(x), obj(y))
    out = np.empty(c.shape, dtype=object)
    for i in np.ndindex(c.shape):
        ci = c[i]
        out[i] = Sym("?", ci, x[i], y[i]) if isinstance(ci, Sym) else (x[i] if ci else y[i])
    return out.view(SymArr)


def sym_minmax(op):
    def f(a, b):
        a, b = np.broadcast_arrays(obj(a), obj(b))
        out = np.empty(a.shape, dtype=object)
        for i in np.ndindex(a.shape):
            out[i] = Sym(op, a[i], b[i]) if isinstance(a[i], Sym) or isinstance(b[i], Sym) else (max if op == "max" else min)(a[i], b[i])
        return out.view(SymArr)
    return f


class SymNP:
    """stand-in for the `np` global of metaworld_amd/policies.py while tracing"""

    def __getattr__(self, k):
        return getattr(np, k)

    where = staticmethod(sym_where)
    maximum, minimum = staticmethod(sym_minmax("max")), staticmethod(sym_minmax("min"))

    @staticmethod
    def asarray(a, dtype=None):
        return a if is_sym(a) else np.asarray(a, dtype=dtype)

    @staticmethod
    def array(a, dtype=None, copy=True):
        return a.copy() if is_sym(a) else np.array(a, dtype=dtype, copy=copy)

    @staticmethod
    def stack(arrs, axis=0):
        return np.stack([obj(a) for a in arrs], axis=axis).view(SymArr) if any(is_sym(a) for a in arrs) else np.stack(arrs, axis=axis)

    @staticmethod
    def broadcast_to(a, shape):
        return np.broadcast_to(a, shape)

    @staticmethod
    def sqrt(a):
        return np.sqrt(a)          # object arrays dispatch to Sym.sqrt


def sym_act(hand, to, p, grab):
    """policies._act: a[:, :3] = p * (to - hand); a[:, 3] = grab  (float64 arithmetic, stored as float32)"""
    xyz = obj(p) * (obj(to) - hand) if not np.isscalar(p) else p * (obj(to) - hand)
    g = np.broadcast_to(obj(grab), (1,))
    return [Sym("f32", v, kind="f") if isinstance(v, Sym) else float(np.float32(v)) for v in list(xyz[0]) + [g[0]]]


def trace(fn):
    Sym.reset()
    obs = np.empty((1, 39), dtype=object).view(SymArr)
    for i in range(39):
        obs[0, i] = Sym("obs", i)
    saved = P.np, P._act
    P.np, P._act = SymNP(), sym_act
    try:
        outs = fn(obs)
    finally:
        P.np, P._act = saved
    return outs


def emit(name, outs):
    live, stack = set(), [o for o in outs if isinstance(o, Sym)]
    while stack:
        s = stack.pop()
        if id(s) in live:
            continue
        live.add(id(s))
        stack += [a for a in s.args if isinstance(a, Sym)]
    lines, n = [], 0
    for s in Sym.order:
        if id(s) not in live:
            continue
        if s.op == "obs":
            s.name = f"o[{s.args[0]}]"
            continue
        s.name = f"t{n}"
        n += 1
        a = [lit(x) for x in s.args]
        ty = {"d": "double", "b": "bool", "f": "float"}[s.kind]
        if s.op in ("+", "-", "*", ">", "<", ">=", "<=", "&&", "||"):
            e = f"{a[0]} {s.op} {a[1]}"
        elif s.op == "neg":
            e = f"-{a[0]}"
        elif s.op == "!":
            e = f"!{a[0]}"
        elif s.op in ("fabs", "sqrt"):
            e = f"{s.op}({a[0]})"
        elif s.op == "?":
            e = f"{a[0]} ? {a[1]} : {a[2]}"
        elif s.op == "max":
            e = f"({a[0]} > {a[1]}) ? {a[0]} : {a[1]}"          # np.maximum / np.minimum on finite values
        elif s.op == "min":
            e = f"({a[0]} < {a[1]}) ? {a[0]} : {a[1]}"
        elif s.op == "f32":
            e = f"(float){a[0]}"
        else:
            raise ValueError(s.op)
        lines.append(f"    const {ty} {s.name} = {e};")
    for k, o in enumerate(outs):
        v = o.name if isinstance(o, Sym) else repr(float(o)) + "f"
        lines.append(f"    a[{k}] = clip1({v});")
    return f"MW_HD void policy_{name}(const double* o, float* a) {{\n    MW_FP_EXACT\n" + "\n".join(lines) + "\n}\n"


def generate():
    parts = ["""// GENERATED by tools/gen_device_policies.py from metaworld_amd/policies.py -- do not edit.
// One straight-line function per v3 task (index = ALL_V3 order = MT50 one-hot id): the scripted policy of
// metaworld/policies/sawyer_*_v3_policy.py as restated in policies.py, same float64 operations in the same order, the action
// stored as float32 and clipped to [-1, 1].  o = the 39 observation values (goal visible), a = the 4 action values.
#pragma once
#include <math.h>

namespace mw {
MW_HD float clip1(float v) { return v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v); }
"""]
    cases = []
    for i, task in enumerate(T.ALL_V3):
        fn = task[:-3].replace("-", "_")
        parts.append(emit(fn, trace(P.POLICIES[task])))
        cases.append(f"        case {i}: policy_{fn}(o, a); break;          // {task}")
    parts.append("MW_HD void scripted_policy(int task_id, const double* o, float* a) {\n    switch (task_id) {\n" + "\n".join(cases)
                 + "\n        default: a[0] = a[1] = a[2] = a[3] = 0.0f;\n    }\n}\n}  // namespace mw\n")
    return "\n".join(parts)


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(text.splitlines())} lines")
