"""Per-operation integer typing: Julia types every integer operation by its operands -- Int32 * Int32 is an
Int32 and wraps at 32 bits, UInt8(1) - UInt8(2) is 255, an integer literal is an Int64, Bool yields to every integer type -- while
the device's integer class computes in ONE 64-bit domain and truncates on store.  The two agree exactly when no value Julia would
have wrapped at a narrower width is observed at a wider one; the planner (csrc/smr_plan.cpp: int_class_matches_julia) admits
exactly those calls and refuses the rest with SMR_EUNSUPPORTED (the binding falls back to the CPU method), and the oracle evaluates
with Julia's typing operation by operation.

Three independent statements of the semantics meet here:
  * `julia_eval` below   -- Julia's promotion rules restated in plain Python integers (the truth of this file),
  * the oracle           -- oracle/strided_oracle.cpp: julia_int_types + eval_prog,
  * `wide_eval` below    -- what the device kernels compute (64-bit wrapping arithmetic, truncation on store; the GPU tests of
                            tests/test_integer_class.py and the fuzzers pin the kernels to it).
Property: oracle == julia_eval always; and whenever the planner ADMITS a call, wide_eval == julia_eval on adversarial values."""
import numpy as np
import pytest

import strided_jl_amd as S
from util import fview, run_oracle

fn = S.fn
DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]


# ---- Julia's semantics on Python integers ------------------------------------------------------------------------------------
def wrap(v, bits, sgn):
    v &= (1 << bits) - 1
    if sgn and v >> (bits - 1):
        v -= 1 << bits
    return v


class Val:
    """(value, bits, signed); bits == 1 is Bool"""

    def __init__(self, v, bits, sgn):
        self.v, self.bits, self.sgn = int(v), bits, sgn


def promote(a, b):
    if a.bits == 1:
        return b.bits, b.sgn
    if b.bits == 1:
        return a.bits, a.sgn
    if a.bits != b.bits:
        return (a.bits, a.sgn) if a.bits > b.bits else (b.bits, b.sgn)
    return a.bits, a.sgn and b.sgn


def julia_eval(tree, args):
    op = tree[0]
    if op == "arg":
        return args[tree[1]]
    if op == "const":
        return Val(tree[1], 64, True)
    xs = [julia_eval(t, args) for t in tree[1:]]
    if op in ("add", "sub", "mul"):
        a, b = xs
        bits, sgn = promote(a, b)
        if a.bits == 1 and b.bits == 1:
            bits, sgn = 64, True
        r = {"add": a.v + b.v, "sub": a.v - b.v, "mul": a.v * b.v}[op]
        return Val(wrap(r, bits, sgn), bits, sgn)
    if op == "neg":
        a, = xs
        bits, sgn = (64, True) if a.bits == 1 else (a.bits, a.sgn)
        return Val(wrap(-a.v, bits, sgn), bits, sgn)
    if op == "abs":
        a, = xs
        return a if a.bits == 1 else Val(wrap(abs(a.v), a.bits, a.sgn), a.bits, a.sgn)
    if op == "abs2":
        a, = xs
        return a if a.bits == 1 else Val(wrap(a.v * a.v, a.bits, a.sgn), a.bits, a.sgn)
    if op in ("min", "max"):
        a, b = xs
        bits, sgn = promote(a, b)
        r = (b.v if b.v < a.v else a.v) if op == "min" else (b.v if a.v < b.v else a.v)
        return Val(r, bits, sgn)
    if op in ("lt", "le", "gt", "ge", "eq", "ne"):
        a, b = xs
        r = {"lt": a.v < b.v, "le": a.v <= b.v, "gt": a.v > b.v, "ge": a.v >= b.v, "eq": a.v == b.v, "ne": a.v != b.v}[op]
        return Val(int(r), 1, False)
    if op == "ifelse":
        c, a, b = xs
        bits, sgn = promote(a, b)
        r = a if c.v != 0 else b
        return Val(r.v, bits, sgn)
    raise ValueError(op)


def wide_eval(tree, args):
    """the device's integer class: everything in wrapping signed 64-bit arithmetic"""
    op = tree[0]
    if op == "arg":
        return args[tree[1]].v
    if op == "const":
        return tree[1]
    xs = [wide_eval(t, args) for t in tree[1:]]
    w = lambda v: wrap(v, 64, True)  # noqa: E731
    if op == "add":
        return w(xs[0] + xs[1])
    if op == "sub":
        return w(xs[0] - xs[1])
    if op == "mul":
        return w(xs[0] * xs[1])
    if op == "neg":
        return w(-xs[0])
    if op == "abs":
        return w(abs(xs[0]))
    if op == "abs2":
        return w(xs[0] * xs[0])
    if op == "min":
        return xs[1] if xs[1] < xs[0] else xs[0]
    if op == "max":
        return xs[1] if xs[0] < xs[1] else xs[0]
    if op in ("lt", "le", "gt", "ge", "eq", "ne"):
        a, b = xs
        return int({"lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b, "eq": a == b, "ne": a != b}[op])
    if op == "ifelse":
        return xs[1] if xs[0] != 0 else xs[2]
    raise ValueError(op)


def to_lambda(tree):
    """the same tree as a Python callable over tracer values (what a user's closure would be)"""
    def build(t, a):
        op = t[0]
        if op == "arg":
            return a[t[1]]
        if op == "const":
            return t[1]
        xs = [build(u, a) for u in t[1:]]
        if op == "neg":
            return -xs[0]
        return getattr(fn, op)(*xs)
    nargs = 1 + max(_args_of(tree))
    return lambda *a: build(tree, a[:nargs]), nargs


def _args_of(t):
    if t[0] == "arg":
        return [t[1]]
    if t[0] == "const":
        return [0]
    out = []
    for u in t[1:]:
        out += _args_of(u)
    return out


def random_tree(rng, depth, nargs):
    if depth == 0 or rng.random() < 0.25:
        if rng.random() < 0.8:
            return ("arg", int(rng.integers(0, nargs)))
        return ("const", [0, 1, 2, 3, -1, -7, 100, 255, 65535, 2 ** 31, -2 ** 31][int(rng.integers(0, 11))])
    r = rng.random()
    if r < 0.55:
        return (str(rng.choice(["add", "sub", "mul"])), random_tree(rng, depth - 1, nargs), random_tree(rng, depth - 1, nargs))
    if r < 0.70:
        return (str(rng.choice(["neg", "abs", "abs2"])), random_tree(rng, depth - 1, nargs))
    if r < 0.82:
        return (str(rng.choice(["min", "max"])), random_tree(rng, depth - 1, nargs), random_tree(rng, depth - 1, nargs))
    if r < 0.93:
        return (str(rng.choice(["lt", "le", "gt", "ge", "eq", "ne"])), random_tree(rng, depth - 1, nargs), random_tree(rng, depth - 1, nargs))
    return ("ifelse", (str(rng.choice(["lt", "ge", "ne"])), random_tree(rng, depth - 1, nargs), random_tree(rng, depth - 1, nargs)),
            random_tree(rng, depth - 1, nargs), random_tree(rng, depth - 1, nargs))


def adversarial(rng, dtype, n):
    info = np.iinfo(dtype)
    edge = [info.min, info.min + 1, -1, 0, 1, 2, info.max - 1, info.max, info.max // 2, info.min // 2, 127, 128, 255, 256]
    edge = [e for e in edge if info.min <= e <= info.max]
    vals = [edge[int(rng.integers(0, len(edge)))] if rng.random() < 0.7 else int(rng.integers(info.min, info.max, endpoint=True, dtype=dtype)) for _ in range(n)]
    return np.asfortranarray(np.array(vals, dtype=dtype))


def planned(f, op, dims, arrays):
    """(admitted, description): does the device planner take the call into its integer class?"""
    try:
        plan = S.make_plan(f, op, None, dims, arrays)
    except S._lib.UnsupportedOnDevice:
        return False, "refused"
    return True, plan.describe()


# ---- the advisor's two probes, with NumPy as a further witness ---------------------------------------------------------------
def test_advisor_probes_follow_julia_and_numpy():
    a, b, c = (np.array([v], dtype=np.uint8) for v in (1, 2, 10))
    out = np.zeros(1, dtype=np.uint8)
    f = lambda x, y, z: fn.min(x - y, z)  # noqa: E731
    got = run_oracle(f, None, None, (1,), (fview(out), fview(a), fview(b), fview(c)))
    assert got[0] == np.minimum(a - b, c)[0] == 10          # UInt8(1) - UInt8(2) == 255, min(255, 10) == 10
    ok, _ = planned(f, None, (1,), (fview(out), fview(a), fview(b), fview(c)))
    assert not ok                                            # an order test on a value Julia wrapped at 8 bits
    x = np.array([2 ** 20], dtype=np.int32)
    wide = np.zeros(1, dtype=np.int64)
    g = lambda u, v: u * v  # noqa: E731
    with np.errstate(over="ignore"):
        want = (x * x).astype(np.int64)                      # 2^40 wraps to 0 in Int32, then widens
    got = run_oracle(g, None, None, (1,), (fview(wide), fview(x), fview(x)))
    assert got[0] == want[0] == 0
    ok, _ = planned(g, None, (1,), (fview(wide), fview(x), fview(x)))
    assert not ok                                            # a 32-bit product observed at 64 bits
    narrow = np.zeros(1, dtype=np.int32)
    ok, desc = planned(g, None, (1,), (fview(narrow), fview(x), fview(x)))
    assert ok and " ct=i64" in desc                          # the same product stored to Int32: congruent modulo 2^32


def test_what_is_admitted_and_what_is_refused():
    rng = np.random.default_rng(0)
    i8, u8, i16, i32, i64 = (fview(adversarial(rng, dt, 8)) for dt in (np.int8, np.uint8, np.int16, np.int32, np.int64))
    u64, u32 = (fview(adversarial(rng, dt, 8)) for dt in (np.uint64, np.uint32))
    sim = lambda v, dt=None: v.similar(dt)  # noqa: E731
    admitted = [
        (lambda a, b: a * b + a, (sim(i32), i32, i32)),                 # ring operations, destination as narrow as the operands
        (lambda a, b: a * b, (sim(i32, np.int16), i32, i32)),           # ... or narrower
        (lambda a: a + 1, (sim(i8, np.int64), i8)),                     # a literal is an Int64: Int8 + 1 is computed at 64 bits
        (lambda a, b: fn.max(a, b), (sim(i16, np.int64), i16, i8)),     # order on untouched values
        (lambda a, b: a < b, (sim(i8, np.uint8), i8, i16)),
        (lambda a, b: (a + 0) * b, (sim(i64), i8, i16)),                # widened to Int64 before anything can wrap
        (lambda a, b: a * b, (sim(i64), i32, i64)),                     # Int32 * Int64 is an Int64
        (lambda a: fn.abs(a), (sim(i8), i8)),                           # abs(typemin(Int8)) == typemin(Int8), same low 8 bits
        (lambda a, b: fn.eq(a, b), (sim(u64, np.uint8), u64, u64)),    # round 5: all-unsigned equality is equality of bit patterns
        (lambda a, b: fn.ne(a, b), (sim(u64, np.uint8), u64, u32)),    # ... UInt32 is zero-extended, as Julia promotes it
    ]
    refused = [
        (lambda a, b: a * b, (sim(i32, np.int64), i32, i32)),           # 32-bit product observed at 64 bits
        (lambda a, b: fn.min(a - b, b), (sim(u8), u8, u8)),             # order on a wrapped difference
        (lambda a, b: (a + b) < b, (sim(i8, np.uint8), i8, i8)),        # comparison of a wrapped sum
        (lambda a: fn.abs(a), (sim(i8, np.int16), i8)),                 # abs(typemin(Int8)) is -128 in Julia, +128 at 64 bits
        (lambda a: -a, (sim(u8, np.int64), u8)),                        # -UInt8(1) == 255
        (lambda a, b: (a < b) + a, (sim(i8, np.int64), i8, i8)),        # Bool + Int8 is an Int8
        (lambda a, b: fn.ifelse(a * a > b, a, b), (sim(i16), i16, i16)),  # the condition observes a wrapped square
        (lambda a, b: fn.eq(a, b), (sim(u64, np.uint8), u64, i64)),    # UInt64 against a signed value: Julia compares mathematically
        (lambda a, b: fn.eq(a - 10, b), (sim(u64, np.uint8), u8, u64)),  # a literal is an Int64: UInt8 - 10 may be negative
        (lambda a, b: a < b, (sim(u64, np.uint8), u64, u64)),           # no order on UInt64 in a signed 64-bit domain
    ]
    for f, arrs in admitted:
        ok, desc = planned(f, None, arrs[0].size, arrs)
        assert ok and " ct=i64" in desc, desc
    for f, arrs in refused:
        ok, desc = planned(f, None, arrs[0].size, arrs)
        assert not ok, desc
    # reductions: the accumulator observes f's value at the destination's width; min / max need it exact
    r64 = S.StridedView(np.zeros(1, dtype=np.int64), i32.size, (0,), 0)
    r32 = S.StridedView(np.zeros(1, dtype=np.int32), i32.size, (0,), 0)
    assert planned(lambda a: a, "+", i32.size, (r64, i32))[0]
    assert planned(fn.abs2, "+", i32.size, (r32, i32))[0]
    assert not planned(fn.abs2, "+", i32.size, (r64, i32))[0]         # abs2(::Int32) wraps at 32 bits before it is summed at 64
    assert not planned(lambda a, b: a - b, "max", i32.size, (r32, i32, i32))[0]
    assert planned(lambda a, b: fn.max(a, b), "max", i32.size, (r32, i32, i32))[0]


# ---- the property ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_admitted_calls_compute_what_julia_computes(seed):
    rng = np.random.default_rng(1000 + seed)
    n, nadmit, nrefuse = 24, 0, 0
    for trial in range(120):
        nargs = int(rng.integers(1, 4))
        tree = random_tree(rng, int(rng.integers(1, 4)), nargs)
        used = sorted(set(_args_of(tree)))
        if tree[0] in ("arg", "const") and trial % 3:
            continue
        dts = [DTYPES[int(rng.integers(0, len(DTYPES)))] for _ in range(nargs)]
        if rng.random() < 0.4:  # homogeneous operands are the common case
            dts = [dts[0]] * nargs
        ddt = DTYPES[int(rng.integers(0, len(DTYPES)))] if rng.random() < 0.6 else dts[0]
        ins = [adversarial(rng, dt, n) for dt in dts]
        f, need = to_lambda(tree)
        views = tuple(fview(a) for a in ins[:need])
        dest = fview(np.zeros(n, dtype=ddt))
        try:
            got = run_oracle(f, None, None, (n,), (dest,) + views)
        except Exception as e:  # noqa: BLE001 -- outside the integer class altogether (e.g. UInt64 under an order test)
            assert "nsupported" in str(e) or "64-bit" in str(e), (tree, e)
            continue
        dbits, dsgn = np.dtype(ddt).itemsize * 8, np.issubdtype(ddt, np.signedinteger)
        jl, wide = [], []
        for i in range(n):
            args = [Val(int(a[i]), a.dtype.itemsize * 8, np.issubdtype(a.dtype, np.signedinteger)) for a in ins[:need]]
            jl.append(wrap(julia_eval(tree, args).v, dbits, dsgn))
            wide.append(wrap(wide_eval(tree, args), dbits, dsgn))
        u64_in = any(dt == np.uint64 for dt in dts[:need]) or ddt == np.uint64
        ordered = any(k in repr(tree) for k in ("min", "max", "lt", "le", "gt", "ge", "eq", "ne", "abs'"))
        if not (u64_in and ordered):  # UInt64 has no order in a signed 64-bit domain: both sides refuse such calls
            assert [int(v) for v in got] == jl, ("oracle vs Julia's typing", tree, dts, ddt)
        ok, desc = planned(f, None, (n,), (dest,) + views)
        if ok and " ct=i64" in desc:
            nadmit += 1
            assert wide == jl, ("the planner admitted a call whose 64-bit evaluation differs from Julia's", tree, dts[:need], ddt, used)
        else:
            nrefuse += 1
    assert nadmit >= 20 and nrefuse >= 10, (nadmit, nrefuse)
