"""Whitelisted elementwise functions usable inside fused expressions -- on `Arg` placeholders when
tracing a map! closure, and on StridedView / Broadcasted operands for dot-fusion (the Python
spelling of `sin.(A)`, `max.(abs.(B1), real.(B3))`, ... from test/othertests.jl:46-66)."""
from __future__ import annotations

import cmath
import math
import numbers

import numpy as np

from .expr import _OpsMixin, as_expr


def _node(op, *args):
    for a in args:
        if isinstance(a, _OpsMixin):
            return a._make(op, *args)
    for a in args:  # StridedView (gets _make from broadcast.py)
        if hasattr(a, "_make"):
            return a._make(op, *args)
    return None


def _unary(op, pyf):
    def g(x):
        n = _node(op, x)
        if n is not None:
            return n
        if isinstance(x, (numbers.Number, np.generic)):
            return pyf(x)
        raise TypeError(f"{op}: unsupported argument {type(x)}")
    g.__name__ = op
    return g


def _cm(realf, cplxf):
    return lambda x: cplxf(x) if isinstance(x, complex) else realf(x)


identity = lambda x: x  # noqa: E731
neg = _unary("neg", lambda x: -x)
abs = _unary("abs", lambda x: x.__abs__())  # noqa: A001
abs2 = _unary("abs2", lambda x: (x * x.conjugate()).real if isinstance(x, complex) else x * x)
conj = _unary("conj", lambda x: x.conjugate())
real = _unary("real", lambda x: x.real)
imag = _unary("imag", lambda x: x.imag)
sqrt = _unary("sqrt", _cm(math.sqrt, cmath.sqrt))
exp = _unary("exp", _cm(math.exp, cmath.exp))
log = _unary("log", _cm(math.log, cmath.log))
sin = _unary("sin", _cm(math.sin, cmath.sin))
cos = _unary("cos", _cm(math.cos, cmath.cos))
tanh = _unary("tanh", _cm(math.tanh, cmath.tanh))
inv = _unary("inv", lambda x: 1 / x)


def _binary(op, pyf):
    def g(x, y, *more):
        n = _node(op, x, y, *more)
        if n is not None:
            return n
        r = pyf(x, y)
        for m in more:
            r = pyf(r, m)
        return r
    g.__name__ = op
    return g


add = _binary("add", lambda a, b: a + b)
sub = _binary("sub", lambda a, b: a - b)
mul = _binary("mul", lambda a, b: a * b)
div = _binary("div", lambda a, b: a / b)
min = _binary("min", lambda a, b: b if b < a else a)  # noqa: A001
max = _binary("max", lambda a, b: b if a < b else a)  # noqa: A001
lt = _binary("lt", lambda a, b: a < b)
le = _binary("le", lambda a, b: a <= b)
gt = _binary("gt", lambda a, b: a > b)
ge = _binary("ge", lambda a, b: a >= b)
eq = _binary("eq", lambda a, b: a == b)
ne = _binary("ne", lambda a, b: a != b)


def ifelse(c, a, b):
    n = _node("select", c, a, b)
    if n is not None:
        return n
    return a if c else b


select = ifelse
__all__ = [k for k in list(globals()) if not k.startswith("_") and k not in
           ("annotations", "cmath", "math", "numbers", "np", "as_expr")]
