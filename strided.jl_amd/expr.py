"""Symbolic form of the fused elementwise function `f` and its serialisation to the f-program
bytecode of include/strided_hip.h.

In the reference `f` is an arbitrary Julia closure: map!'s argument (src/mapreduce.jl:38-53) or
the `CaptureArgs` functor tree that broadcast lowering builds (src/broadcast.jl:67-98), whose
`Arg()` leaves consume the array values positionally, depth first, and whose inner nodes apply
`bc.f` left to right.  A closure cannot cross a C ABI, so the host side traces it: `f` is called
once with `Arg` placeholders and the resulting tree of whitelisted operations is emitted as
postfix code.  Anything outside the whitelist raises (the Julia shim would fall back to the CPU
method).
"""
from __future__ import annotations

import numbers

import numpy as np

from ._lib import OPCODES, SMR_MAXCONST, SMR_MAXPROG

_UNARY = {"neg": "NEG", "abs": "ABS", "abs2": "ABS2", "conj": "CONJ", "real": "REAL", "imag": "IMAG",
          "sqrt": "SQRT", "exp": "EXP", "log": "LOG", "sin": "SIN", "cos": "COS", "tanh": "TANH",
          "inv": "INV"}
_BINARY = {"add": "ADD", "sub": "SUB", "mul": "MUL", "div": "DIV", "min": "MIN", "max": "MAX",
           "lt": "LT", "le": "LE", "gt": "GT", "ge": "GE", "eq": "EQ", "ne": "NE"}
_NARY_FOLD = {"add", "mul", "min", "max"}  # Julia's +(a,b,c,d) folds left: ((a+b)+c)+d


class _OpsMixin:
    """Operator overloading shared by Expr (map! closures) and Broadcasted (dot-fusion)."""

    # NumPy scalars on the LEFT (`np.float32(0.5) * a`) must reach __rmul__ & co. as they are: without this NumPy converts the
    # scalar to a Python float first and a Float32 literal would silently become a Float64 one (and widen the whole call)
    __array_ufunc__ = None

    @classmethod
    def _make(cls, op, *args):  # overridden
        raise NotImplementedError

    def __add__(self, o): return self._make("add", self, o)
    def __radd__(self, o): return self._make("add", o, self)
    def __sub__(self, o): return self._make("sub", self, o)
    def __rsub__(self, o): return self._make("sub", o, self)
    def __mul__(self, o): return self._make("mul", self, o)
    def __rmul__(self, o): return self._make("mul", o, self)
    def __truediv__(self, o): return self._make("div", self, o)
    def __rtruediv__(self, o): return self._make("div", o, self)
    def __neg__(self): return self._make("neg", self)
    def __pos__(self): return self
    def __abs__(self): return self._make("abs", self)
    def __lt__(self, o): return self._make("lt", self, o)
    def __le__(self, o): return self._make("le", self, o)
    def __gt__(self, o): return self._make("gt", self, o)
    def __ge__(self, o): return self._make("ge", self, o)
    # == / != stay Python identity semantics; use fn.eq / fn.ne for elementwise comparison
    __hash__ = object.__hash__


class Expr(_OpsMixin):
    @classmethod
    def _make(cls, op, *args):
        return Call(op, tuple(as_expr(a) for a in args))


class Arg(Expr):
    """Placeholder for the k-th array argument (1-based), `Arg()` of src/broadcast.jl:71-72."""

    def __init__(self, index: int):
        self.index = int(index)

    def __repr__(self):
        return f"a{self.index}"


class Const(Expr):
    """A captured scalar (src/broadcast.jl:81-83: Ref / 0-dim args are unwrapped, other
    non-array args captured as they are)."""

    def __init__(self, value):
        if isinstance(value, (np.generic,)):
            self.dtype = value.dtype           # strongly typed, like a Julia Float32 literal
            value = value.item()
        elif isinstance(value, bool):
            self.dtype = None                  # weak (Julia Bool/Int do not widen floats)
        elif isinstance(value, numbers.Integral):
            self.dtype = None
            # constants travel as Float64: an integer beyond 2^53 must survive that exactly -- or be one of the type
            # limits the integer class saturates to (typemax(Int64) / typemin(Int64), the seeds of min / max)
            iv = int(value)
            if abs(iv) > 2 ** 53 and int(float(iv)) != iv and iv not in (2 ** 63 - 1, -2 ** 63):
                raise OverflowError(f"integer constant {iv} is not representable in the f-program (Float64 constants)")
        elif isinstance(value, numbers.Rational):
            self.dtype = None                  # Fraction ~ Julia Rational: takes the array's float type
            value = value.numerator / value.denominator
        elif isinstance(value, numbers.Real):
            self.dtype = np.dtype(np.float64)  # a Python float is Julia's Float64
        elif isinstance(value, numbers.Complex):
            self.dtype = np.dtype(np.complex128)
        elif hasattr(value, "numerator") and hasattr(value, "denominator"):
            self.dtype = None                  # Fraction ~ Julia Rational: takes the array's float type
            value = value.numerator / value.denominator
        else:
            raise TypeError(f"cannot capture {type(value)} as a broadcast constant")
        self.value = complex(value)

    def __repr__(self):
        return f"{self.value.real:g}" if self.value.imag == 0 else f"{self.value:g}"


class Call(Expr):
    def __init__(self, op: str, args: tuple):
        if op not in _UNARY and op not in _BINARY and op != "select":
            raise NotImplementedError(f"operation {op!r} is outside the device whitelist")
        self.op, self.args = op, args

    def __repr__(self):
        return f"{self.op}({', '.join(map(repr, self.args))})"


def as_expr(x) -> Expr:
    if isinstance(x, Expr):
        return x
    if hasattr(x, "numerator") or isinstance(x, (numbers.Number, np.generic)):
        return Const(x)
    raise TypeError(f"cannot use {type(x)} inside a fused elementwise function")


# ---- tracing --------------------------------------------------------------------------------------
def trace(f, nargs: int) -> Expr:
    """Turn map!'s `f` into an Expr by calling it on placeholders.  `f` may already be an Expr
    (e.g. a CaptureArgs tree from broadcast lowering) or one of a few names."""
    if isinstance(f, Expr):
        return f
    if isinstance(f, str):
        from . import fn
        f = getattr(fn, f)
    out = f(*[Arg(i + 1) for i in range(nargs)])
    return as_expr(out)


# ---- serialisation ---------------------------------------------------------------------------------
def node_dtype(n, arg_dtypes):
    """Element type Julia would give this sub-expression (None = weakly typed constant)."""
    if isinstance(n, Arg):
        return np.dtype(arg_dtypes[n.index - 1])
    if isinstance(n, Const):
        return n.dtype
    ts = [node_dtype(a, arg_dtypes) for a in n.args]
    strong = [t for t in ts if t is not None]
    if n.op in ("lt", "le", "gt", "ge", "eq", "ne"):
        return np.dtype(np.bool_)
    if n.op in ("abs", "abs2", "real", "imag"):
        return _real_of(strong[0]) if strong else None
    if n.op in ("sqrt", "exp", "log", "sin", "cos", "tanh", "inv"):
        return _float_of(strong[0]) if strong else np.dtype(np.float64)
    if n.op == "select":
        strong = [t for t in ts[1:] if t is not None]
    if not strong:
        return None
    r = np.result_type(*strong)
    if n.op == "div" and not np.issubdtype(r, np.inexact):
        r = np.dtype(np.float64)
    if r == np.bool_ and n.op in ("add", "sub", "mul"):
        r = np.dtype(np.int64)
    return np.dtype(r)


def serialize(e: Expr, arg_dtypes=None, wide: bool = False):
    """Expr -> (code bytes, constants as (re, im) list).  Post-order; leaves are visited left to
    right, which is the order `consume` pops array values in (src/broadcast.jl:86-98).

    Per-operation typing: Julia types every operation of the fused expression separately
    (`Float32 .* Float32` is a Float32 product even when the result is later widened), the device
    computes a whole call in one class.  When the call computes in Float64 (`wide`) although an
    operation's Julia type is Float32 / ComplexF32, a ROUND32 follows it: for + - * / sqrt the
    double-rounded result equals the Float32 operation exactly (53 >= 2*24 + 2 bits)."""
    code, consts = [], []
    narrow = (np.dtype(np.float32), np.dtype(np.complex64))

    def rounds(node) -> bool:
        if not wide or arg_dtypes is None or not isinstance(node, Call):
            return False
        if node.op == "select":
            return False  # picks one of two already rounded values
        return node_dtype(node, arg_dtypes) in narrow

    def const_index(v: complex) -> int:
        for i, c in enumerate(consts):
            if c == v and np.signbit(c.real) == np.signbit(v.real) and np.signbit(c.imag) == np.signbit(v.imag):
                return i
        consts.append(v)
        if len(consts) > SMR_MAXCONST:
            raise NotImplementedError("too many captured constants for the device f-program")
        return len(consts) - 1

    def emit(node):
        if isinstance(node, Arg):
            code.extend((OPCODES["ARG"], node.index))
        elif isinstance(node, Const):
            code.extend((OPCODES["CONST"], const_index(node.value)))
        elif isinstance(node, Call):
            if node.op in _UNARY:
                emit(node.args[0])
                code.extend((OPCODES[_UNARY[node.op]], 0))
                if rounds(node):
                    code.extend((OPCODES["ROUND32"], 0))
            elif node.op == "select":
                for a in node.args:
                    emit(a)
                code.extend((OPCODES["SELECT"], 0))
            else:
                if len(node.args) < 2 or (len(node.args) > 2 and node.op not in _NARY_FOLD):
                    raise ValueError(f"{node.op} takes two arguments")
                emit(node.args[0])
                for a in node.args[1:]:
                    emit(a)
                    code.extend((OPCODES[_BINARY[node.op]], 0))
                    if rounds(node):  # every step of a left fold has the fold's type
                        code.extend((OPCODES["ROUND32"], 0))
        else:
            raise TypeError(type(node))

    emit(e)
    if wide and arg_dtypes is not None and not any(
            np.dtype(t) in (np.dtype(np.float64), np.dtype(np.complex128)) or np.issubdtype(np.dtype(t), np.integer) or np.dtype(t) == np.bool_
            for t in arg_dtypes):
        # the 64-bit class comes from a scalar, not from an operand: tell the library (SMR_OP_WIDEN)
        code.extend((OPCODES["WIDEN"], 0))
    if len(code) // 2 > SMR_MAXPROG:
        raise NotImplementedError("fused expression too long for the device f-program")
    return bytes(code), consts


def needs_wide(e: Expr, arg_dtypes) -> bool:
    """Does Julia evaluate some part of f in a 64-bit (or integer) type?  True for Float64 / ComplexF64 /
    integer arrays, and for strongly typed 64-bit scalars (`A32 .* 0.1`: a Float64 literal)."""
    wide = (np.dtype(np.float64), np.dtype(np.complex128))

    def walk(n):
        t = node_dtype(n, arg_dtypes)
        if t is not None and (t in wide or np.issubdtype(t, np.integer)):
            return True
        return isinstance(n, Call) and any(walk(a) for a in n.args)

    return walk(e)


def max_arg(e: Expr) -> int:
    if isinstance(e, Arg):
        return e.index
    if isinstance(e, Call):
        return max((max_arg(a) for a in e.args), default=0)
    return 0


# ---- result element type (Broadcast.combine_eltypes analogue) ---------------------------------------
def _real_of(dt):
    dt = np.dtype(dt)
    if dt == np.complex64:
        return np.dtype(np.float32)
    if dt == np.complex128:
        return np.dtype(np.float64)
    return dt


def _float_of(dt):
    dt = np.dtype(dt)
    if np.issubdtype(dt, np.inexact):
        return dt
    return np.dtype(np.float64)


def result_dtype(e: Expr, arg_dtypes):
    """Element type Julia would infer for f(args...): weak (Int/Bool/Rational) constants do not
    widen, a Python float is a Float64, comparisons give Bool."""

    r = node_dtype(e, arg_dtypes)
    return np.dtype(np.float64) if r is None else r
