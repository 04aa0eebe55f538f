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
from strided_jl_amd.stridedview import smr_dtype  # noqa: E402

I64 = smr_dtype(np.int64)
DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]
INPUTS = DTYPES + [np.bool_, np.bool_]   # operands may be Bool arrays (SMR_BOOL); destinations of arithmetic are integers


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
        if a.bits == 1 and b.bits == 1 and op != "mul":   # Bool (+,-) Bool is an Int; Bool * Bool stays a Bool
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


# ---- the device's integer class on Python integers ----------------------------------------------------------------------------
WRAPS = {23: (8, True), 24: (16, True), 25: (32, True), 26: (8, False), 27: (16, False), 28: (32, False)}   # SMR_OP_WRAP_*
OPN = {v: k for k, v in S._lib.OPCODES.items()}


def canon_prog(f, op, dims, arrays):
    """the program the kernels would run: (code pairs, constants, wraps added, caller's index of every canonical operand), or None if refused"""
    import ctypes as C
    p, keep = S.build_problem(f, op, None, dims, arrays, stream=0)
    lib = S._lib.load()
    code = (C.c_uint8 * (2 * S._lib.SMR_MAXPROG))()
    nw, ct, orig = C.c_int(0), C.c_int(0), (C.c_int32 * 8)()
    n = lib.smr_debug_canon_prog(C.byref(p), code, len(code), C.byref(nw), C.byref(ct), orig)
    if n < 0:
        assert n == S._lib.SMR_EUNSUPPORTED, n
        return None
    consts = [p.fconsts[2 * i] for i in range(p.nconsts)]
    return [(code[2 * i], code[2 * i + 1]) for i in range(n)], consts, nw.value, list(orig), ct.value


def wide_eval(prog, consts, orig, argvals):
    """everything in wrapping signed 64-bit arithmetic (csrc/smr_device.h: mathx<ix64>); argvals[k] = value of the caller's operand k"""
    w = lambda v: wrap(v, 64, True)  # noqa: E731
    st = []
    for op, imm in prog:
        name = OPN.get(op)
        if name == "ARG":
            st.append(w(argvals[orig[imm]]))
        elif name == "CONST":
            st.append(w(int(consts[imm])))
        elif op in WRAPS:
            st[-1] = wrap(st[-1], *WRAPS[op])
        elif name == "NEG":
            st[-1] = w(-st[-1])
        elif name == "ABS":
            st[-1] = w(abs(st[-1]))
        elif name == "ABS2":
            st[-1] = w(st[-1] * st[-1])
        elif name in ("ADD", "SUB", "MUL", "MIN", "MAX", "LT", "LE", "GT", "GE", "EQ", "NE"):
            y = st.pop()
            x = st.pop()
            st.append({"ADD": lambda: w(x + y), "SUB": lambda: w(x - y), "MUL": lambda: w(x * y), "MIN": lambda: y if y < x else x,
                       "MAX": lambda: y if x < y else x, "LT": lambda: int(x < y), "LE": lambda: int(x <= y), "GT": lambda: int(x > y),
                       "GE": lambda: int(x >= y), "EQ": lambda: int(x == y), "NE": lambda: int(x != y)}[name]())
        elif name == "SELECT":
            e = st.pop()
            t = st.pop()
            c = st.pop()
            st.append(t if c != 0 else e)
        else:
            raise ValueError((op, name))
    assert len(st) == 1
    return st[0]


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


def const_only(t):
    return t[0] == "const" or (t[0] != "arg" and all(const_only(u) for u in t[1:]))


def folds(t):
    """an operation on literals only: Python evaluates it (in unbounded integers, comparisons to True / False) before the tracer sees
    anything, so the host passes one Int64 literal where Julia has a wrapped Int64 or a Bool (UInt16 - true is a UInt16, UInt16 - 1 an
    Int64) -- not a program the mirror can express"""
    if t[0] in ("arg", "const"):
        return False
    return const_only(t) or any(folds(u) for u in t[1:])


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
    if dtype == np.bool_:
        return np.asfortranarray(rng.integers(0, 2, size=n).astype(np.bool_))
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
def wraps_of(f, op, dims, arrays):
    """(admitted into the integer class, SMR_OP_WRAP_* instructions the planner added)"""
    cp = canon_prog(f, op, dims, arrays)
    if cp is None:
        return False, 0
    return cp[4] == I64, cp[2]


def test_advisor_probes_follow_julia_and_numpy():
    a, b, c = (np.array([v], dtype=np.uint8) for v in (1, 2, 10))
    out = np.zeros(1, dtype=np.uint8)
    f = lambda x, y, z: fn.min(x - y, z)  # noqa: E731
    got = run_oracle(f, None, None, (1,), (fview(out), fview(a), fview(b), fview(c)))
    assert got[0] == np.minimum(a - b, c)[0] == 10          # UInt8(1) - UInt8(2) == 255, min(255, 10) == 10
    arrs = (fview(out), fview(a), fview(b), fview(c))
    assert wraps_of(f, None, (1,), arrs) == (True, 1)        # an order test on a value Julia wrapped at 8 bits: re-wrapped before the min
    prog, consts, nw, orig, _ = canon_prog(f, None, (1,), arrs)
    assert [o for o, _ in prog] == [0, 0, 33, 26, 0, 36]     # a b SUB WRAP_U8 c MIN
    assert wide_eval(prog, consts, orig, [0, 1, 2, 10]) == 10
    x = np.array([2 ** 20], dtype=np.int32)
    wide = np.zeros(1, dtype=np.int64)
    g = lambda u, v: u * v  # noqa: E731
    with np.errstate(over="ignore"):
        want = (x * x).astype(np.int64)                      # 2^40 wraps to 0 in Int32, then widens
    got = run_oracle(g, None, None, (1,), (fview(wide), fview(x), fview(x)))
    assert got[0] == want[0] == 0
    assert wraps_of(g, None, (1,), (fview(wide), fview(x), fview(x))) == (True, 1)   # a 32-bit product observed at 64 bits: one wrap at the end
    narrow = np.zeros(1, dtype=np.int32)
    ok, desc = planned(g, None, (1,), (fview(narrow), fview(x), fview(x)))
    assert ok and " ct=i64" in desc and "int_wraps" not in desc   # the same product stored to Int32: congruent modulo 2^32, nothing added


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
    rewrapped = [   # (f, arrays, wraps added): before round 5 the planner refused these
        (lambda a, b: a * b, (sim(i32, np.int64), i32, i32), 1),           # 32-bit product observed at 64 bits
        (lambda a, b: fn.min(a - b, b), (sim(u8), u8, u8), 1),             # order on a wrapped difference
        (lambda a, b: (a + b) < b, (sim(i8, np.uint8), i8, i8), 1),        # comparison of a wrapped sum
        (lambda a: fn.abs(a), (sim(i8, np.int16), i8), 1),                 # abs(typemin(Int8)) is -128 in Julia, +128 at 64 bits
        (lambda a: -a, (sim(u8, np.int64), u8), 1),                        # -UInt8(1) == 255
        (lambda a, b: (a < b) + a, (sim(i8, np.int64), i8, i8), 1),        # Bool + Int8 is an Int8
        (lambda a, b: fn.ifelse(a * a > b, a, b), (sim(i16), i16, i16), 1),  # the condition observes a wrapped square
        (lambda a, b, c: (a * b + c) * a, (sim(i32), i16, i16, i32), 1),   # Int16 product, converted to Int32 before the sum: exact first
        (lambda a, b, c: (a * b + c) * a, (sim(i32), i32, i32, i32), 0),   # ... all Int32: a ring at one width, nothing to do
        (lambda a, b: fn.max(a * b, a) - fn.abs(b - a), (sim(i64), i8, i8), 3),
    ]
    refused = [
        (lambda a, b: fn.eq(a, b), (sim(u64, np.uint8), u64, i64)),    # UInt64 against a signed value: Julia compares mathematically
        (lambda a, b: fn.eq(a - 10, b), (sim(u64, np.uint8), u8, u64)),  # a literal is an Int64: UInt8 - 10 may be negative
        (lambda a, b: a < b, (sim(u64, np.uint8), u64, u64)),           # no order on UInt64 in a signed 64-bit domain
        (lambda a, b: fn.ifelse(a < b, a, b) + a, (sim(i64), i8, i32)),  # ifelse(::Bool, ::Int8, ::Int32) + Int8: the sum's type depends on the data
    ]
    for f, arrs, nw in rewrapped:
        assert wraps_of(f, None, arrs[0].size, arrs) == (True, nw), (nw, canon_prog(f, None, arrs[0].size, arrs))
    for f, arrs in admitted:
        ok, desc = planned(f, None, arrs[0].size, arrs)
        assert ok and " ct=i64" in desc and "int_wraps" not in desc, desc
    for f, arrs in refused:
        ok, desc = planned(f, None, arrs[0].size, arrs)
        assert not ok, desc
    # reductions: the accumulator observes f's value at the destination's width; min / max need it exact
    r64 = S.StridedView(np.zeros(1, dtype=np.int64), i32.size, (0,), 0)
    r32 = S.StridedView(np.zeros(1, dtype=np.int32), i32.size, (0,), 0)
    assert wraps_of(lambda a: a, "+", i32.size, (r64, i32)) == (True, 0)
    assert wraps_of(fn.abs2, "+", i32.size, (r32, i32)) == (True, 0)
    assert wraps_of(fn.abs2, "+", i32.size, (r64, i32)) == (True, 1)   # abs2(::Int32) wraps at 32 bits before it is summed at 64
    assert wraps_of(lambda a, b: a - b, "max", i32.size, (r32, i32, i32)) == (True, 1)
    assert wraps_of(lambda a, b: fn.max(a, b), "max", i32.size, (r32, i32, i32)) == (True, 0)


# ---- the property ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(60))
def test_admitted_calls_compute_what_julia_computes(seed):
    rng = np.random.default_rng(1000 + seed)
    n, nadmit, nrefuse, nwrapped = 24, 0, 0, 0
    for trial in range(120):
        nargs = int(rng.integers(1, 4))
        tree = random_tree(rng, int(rng.integers(1, 4)), nargs)
        used = sorted(set(_args_of(tree)))
        if folds(tree):
            continue
        if tree[0] in ("arg", "const") and trial % 3:
            continue
        dts = [INPUTS[int(rng.integers(0, len(INPUTS)))] for _ in range(nargs)]
        if rng.random() < 0.4:  # homogeneous operands are the common case
            dts = [dts[0]] * nargs
        ddt = DTYPES[int(rng.integers(0, len(DTYPES)))] if rng.random() < 0.6 or dts[0] == np.bool_ else dts[0]
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
            args = [Val(int(a[i]), 1 if a.dtype == np.bool_ else a.dtype.itemsize * 8, np.issubdtype(a.dtype, np.signedinteger)) for a in ins[:need]]
            jl.append(wrap(julia_eval(tree, args).v, dbits, dsgn))
        u64_in = any(dt == np.uint64 for dt in dts[:need]) or ddt == np.uint64
        ordered = any(k in repr(tree) for k in ("min", "max", "lt", "le", "gt", "ge", "eq", "ne", "abs'"))
        if not (u64_in and ordered):  # UInt64 has no order in a signed 64-bit domain: both sides refuse such calls
            assert [int(v) for v in got] == jl, ("oracle vs Julia's typing", tree, dts, ddt)
        cp = canon_prog(f, None, (n,), (dest,) + views)
        if cp is not None and cp[4] == I64:
            prog, consts, nw, orig, _ = cp
            nadmit += 1
            nwrapped += nw > 0
            for i in range(n):
                vals = [0] + [int(a[i]) for a in ins[:need]]
                wide.append(wrap(wide_eval(prog, consts, orig, vals), dbits, dsgn))
            assert wide == jl, ("the planner admitted a call whose 64-bit evaluation differs from Julia's", tree, dts[:need], ddt, used, prog)
        else:
            nrefuse += 1
    assert nadmit >= 40 and nwrapped >= 5, (nadmit, nwrapped, nrefuse)


@pytest.mark.parametrize("seed", range(30))
def test_admitted_reductions_compute_what_julia_computes(seed):
    """The same property for reductions into an Int64 accumulator (sum / maximum / minimum of f over the elements): the reduction
    observes f's value at 64 bits, so a narrow result must be re-wrapped before it is accumulated -- abs2(::Int32) summed at 64 bits is
    the sum of the WRAPPED squares."""
    rng = np.random.default_rng(5000 + seed)
    n, nadmit, nwrapped = 24, 0, 0
    for trial in range(80):
        nargs = int(rng.integers(1, 3))
        tree = random_tree(rng, int(rng.integers(1, 4)), nargs)
        if folds(tree) or tree[0] == "const":
            continue
        ins_dt = [dt for dt in INPUTS if dt != np.uint64]          # UInt64 + Int64 promotes to UInt64: the accumulator would not be an Int64
        dts = [ins_dt[int(rng.integers(0, len(ins_dt)))] for _ in range(nargs)]
        op = ["+", "max", "min"][int(rng.integers(0, 3))]
        ins = [adversarial(rng, dt, n) for dt in dts]
        f, need = to_lambda(tree)
        views = tuple(fview(a) for a in ins[:need])
        d0 = int(rng.integers(-1000, 1000))
        acc = np.array([d0], dtype=np.int64)
        dest = S.StridedView(acc, (n,), (0,), 0)
        try:
            got = int(run_oracle(f, op, None, (n,), (dest,) + views).ravel()[0])
        except Exception as e:  # noqa: BLE001
            assert "nsupported" in str(e) or "64-bit" in str(e), (tree, e)
            continue
        vals = []
        for i in range(n):
            args = [Val(int(a[i]), 1 if a.dtype == np.bool_ else a.dtype.itemsize * 8, np.issubdtype(a.dtype, np.signedinteger)) for a in ins[:need]]
            vals.append(julia_eval(tree, args).v)
        fold = {"+": lambda xs: wrap(d0 + sum(xs), 64, True), "max": lambda xs: max([d0] + xs), "min": lambda xs: min([d0] + xs)}[op]
        assert got == fold(vals), ("oracle vs Julia's typing", tree, dts[:need], op)
        cp = canon_prog(f, op, (n,), (S.StridedView(np.array([d0], dtype=np.int64), (n,), (0,), 0),) + views)
        if cp is None or cp[4] != I64:
            continue
        prog, consts, nw, orig, _ = cp
        nadmit += 1
        nwrapped += nw > 0
        wide = [wide_eval(prog, consts, orig, [0] + [int(a[i]) for a in ins[:need]]) for i in range(n)]
        assert fold(wide) == got, ("the planner admitted a reduction whose 64-bit evaluation differs from Julia's", tree, dts[:need], op, prog)
    assert nadmit >= 25 and nwrapped >= 5, (nadmit, nwrapped)


# ---- on the device --------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("jit", [1, 0])
def test_hip_rewrapped_programs_equal_the_oracle(jit):
    """Round 5: programs the integer class used to refuse now run with SMR_OP_WRAP_* instructions where Julia's narrow types are
    observed -- compiled functors (hiprtc) and the interpreter, bit for bit against the oracle (which types every operation like
    Julia) and, for the two advisor probes, against NumPy's fixed-width arithmetic."""
    from util import run_device
    rng = np.random.default_rng(77)
    n = 4096
    old = S.get_option("jit")
    S.set_option("jit", old if jit else 0)
    try:
        x, y = adversarial(rng, np.int32, n), adversarial(rng, np.int32, n)
        got = run_device(lambda u, v: u * v, None, None, (n,), (fview(np.zeros(n, dtype=np.int64)), fview(x), fview(y)))
        with np.errstate(over="ignore"):
            assert np.array_equal(got, (x * y).astype(np.int64))
        a, b, c = (adversarial(rng, np.uint8, n) for _ in range(3))
        got = run_device(lambda p, q, r: fn.min(p - q, r), None, None, (n,), (fview(np.zeros(n, dtype=np.uint8)), fview(a), fview(b), fview(c)))
        assert np.array_equal(got, np.minimum(a - b, c))
        r64 = np.zeros(1, dtype=np.int64)
        got = run_device(fn.abs2, "+", None, (n,), (S.StridedView(r64, (n,), (0,), 0), fview(x)))
        with np.errstate(over="ignore"):
            assert int(np.asarray(got).ravel()[0]) == int((x * x).astype(np.int64).sum())    # abs2(::Int32) wraps at 32 bits, the sum runs at 64
        m1, m2 = adversarial(rng, np.bool_, n), adversarial(rng, np.bool_, n)    # SMR_BOOL: -true == -1, true + true is an Int
        i8 = adversarial(rng, np.int8, n)
        got = run_device(lambda p, q, x: (p + q) * 200 - p + (q + x), None, None, (n,), (fview(np.zeros(n, dtype=np.int64)), fview(m1), fview(m2), fview(i8)))
        with np.errstate(over="ignore"):
            want = (m1.astype(np.int64) + m2) * 200 - m1 + (q8 := (i8 + m2.astype(np.int8))).astype(np.int64)
        assert np.array_equal(got, want) and int((q8 == -128).sum()) > 0
        nrun = nwrapped = 0
        for trial in range(400 if jit == 0 else 120):
            nargs = int(rng.integers(1, 4))
            tree = random_tree(rng, int(rng.integers(1, 4)), nargs)
            if folds(tree) or tree[0] in ("arg", "const"):
                continue
            dts = [INPUTS[int(rng.integers(0, len(INPUTS)))] for _ in range(nargs)]
            ddt = DTYPES[int(rng.integers(0, len(DTYPES)))]
            ins = [adversarial(rng, dt, 192) for dt in dts]
            f, need = to_lambda(tree)
            views = tuple(fview(v) for v in ins[:need])
            dest = fview(np.zeros(192, dtype=ddt))
            cp = canon_prog(f, None, (192,), (dest,) + views)
            if cp is None or cp[4] != I64 or (jit and cp[2] == 0):     # (compiled functors: only the re-wrapped programs, hiprtc takes ~0.3 s each)
                continue
            got = run_device(f, None, None, (192,), (fview(np.zeros(192, dtype=ddt)),) + views)
            want = run_oracle(f, None, None, (192,), (dest,) + views)
            assert np.array_equal(got, want), (tree, dts[:need], ddt, cp[0])
            nrun += 1
            nwrapped += cp[2] > 0
        assert nwrapped >= (10 if jit else 25), (nrun, nwrapped)
        nred = 0
        for trial in range(0 if jit else 150):                          # reductions into an Int64 accumulator (interpreter only: no hiprtc time)
            nargs = int(rng.integers(1, 3))
            tree = random_tree(rng, int(rng.integers(1, 4)), nargs)
            if folds(tree) or tree[0] == "const":
                continue
            ins_dt = [dt for dt in INPUTS if dt != np.uint64]
            ins = [adversarial(rng, ins_dt[int(rng.integers(0, len(ins_dt)))], 3000) for _ in range(nargs)]
            op = ["+", "max", "min"][int(rng.integers(0, 3))]
            f, need = to_lambda(tree)
            views = tuple(fview(v) for v in ins[:need])
            mk = lambda: S.StridedView(np.array([17], dtype=np.int64), (3000,), (0,), 0)  # noqa: E731
            cp = canon_prog(f, op, (3000,), (mk(),) + views)
            if cp is None or cp[4] != I64 or cp[2] == 0:
                continue
            got = run_device(f, op, None, (3000,), (mk(),) + views)
            want = run_oracle(f, op, None, (3000,), (mk(),) + views)
            assert int(np.asarray(got).ravel()[0]) == int(np.asarray(want).ravel()[0]), (tree, op, [v.dtype for v in ins[:need]], cp[0])
            nred += 1
        assert jit or nred >= 15, nred
    finally:
        S.set_option("jit", old)


def test_bool_operands_are_typed_like_julia():
    """SMR_BOOL (round 5; before, Bool arrays were passed as UInt8): Bool yields to every integer type, -true and true + true are Ints."""
    m1 = np.array([True, True, False, True])
    m2 = np.array([True, False, False, True])
    a = np.array([127, -128, 5, -1], dtype=np.int8)
    out = lambda dt=np.int64: fview(np.zeros(4, dtype=dt))  # noqa: E731
    cases = [
        (lambda p: -p, (m1,), [-1, -1, 0, -1], 0),                                   # -true == -1 (a UInt8 would give 255)
        (lambda p, q: (p + q) * 200, (m1, m2), [400, 200, 0, 400], 0),                # true + true is an Int (a UInt8 sum times 200 would wrap to 144)
        (lambda p, x: p + x, (m1, a), [-128, -127, 5, 0], 1),                         # Bool + Int8 is an Int8: 127 + true wraps to -128
        (lambda p, q: p * q, (m1, m2), [1, 0, 0, 1], 0),                              # Bool * Bool stays a Bool
        (lambda p, x: fn.ifelse(p, x, -x), (m1, a), [127, -128, -5, -1], 1),          # -Int8(-128) == -128 observed at 64 bits
        (lambda p, x: p * x * x, (m1, a), [1, 0, 0, 1], 1),                           # Int8 products wrap at 8 bits: 127 * 127 == 1
    ]
    for f, ins, want, nw in cases:
        views = tuple(fview(np.asfortranarray(v)) for v in ins)
        got = run_oracle(f, None, None, (4,), (out(),) + views)
        assert [int(v) for v in got] == want, (want, got)
        cp = canon_prog(f, None, (4,), (out(),) + views)
        assert cp is not None and cp[4] == I64 and cp[2] == nw, cp
        prog, consts, _, orig, _ = cp
        for i in range(4):
            assert wide_eval(prog, consts, orig, [0] + [int(v[i]) for v in ins]) == want[i]
    # a Bool array still moves as bytes and counts as before
    r = S.StridedView(np.zeros(1, dtype=np.int64), (4,), (0,), 0)
    assert int(run_oracle(lambda p: p, "+", None, (4,), (r, fview(m1)))[0]) == 3
    plan = S.make_plan(lambda p: p, None, None, (4,), (fview(np.zeros(4, dtype=np.bool_)), fview(m1)))
    assert "(bitcopy)" in plan.describe()
