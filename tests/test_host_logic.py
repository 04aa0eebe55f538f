"""Host-side mirror of the reference's interface: StridedView algebra (StridedViews.jl
semantics), broadcast lowering (src/broadcast.jl), f-program serialisation, result-eltype
inference, argument errors raised before the funnel (src/mapreduce.jl:38-53,74-96)."""
from fractions import Fraction

import numpy as np
import pytest

import strided_jl_amd as S
from strided_jl_amd import expr as E
from strided_jl_amd import fn
from strided_jl_amd._lib import OPCODES as OP


def F(a):
    return S.StridedView(np.asfortranarray(a))


def test_lazy_view_algebra_matches_numpy():
    rng = np.random.default_rng(0)
    a0 = np.asfortranarray(rng.standard_normal((4, 5, 6)) + 1j * rng.standard_normal((4, 5, 6)))
    a = F(a0)
    assert a.strides == (1, 4, 20) and a.offset == 0
    assert np.array_equal(a.permutedims((2, 0, 1)).toarray(), a0.transpose(2, 0, 1))
    assert np.array_equal(a.conj().toarray(), a0.conj())
    m = F(a0[:, :, 0])
    assert m.adjoint().op == "conj" and m.adjoint().strides == (4, 1)
    assert np.array_equal(m.adjoint().toarray(), a0[:, :, 0].conj().T)
    assert m.adjoint().adjoint().op == "identity"
    v = a[1:4:2, ::-1, 5]
    assert v.size == (2, 5) and v.strides == (2, -4) and np.array_equal(v.toarray(), a0[1:4:2, ::-1, 5])
    assert np.array_equal(a[3:4, 2:3, :].toarray(), a0[3:4, 2:3, :])
    assert a[1, 2, 3] == a0[1, 2, 3]
    # real views ignore conj (Base.conj!(a::StridedView{<:Real}) = a)
    r = F(np.zeros((3, 3)))
    assert r.conj().op == "identity" and r.adjoint().op == "identity"


def test_sreshape_only_merges_contiguous_dims():
    a0 = np.asfortranarray(np.arange(10 * 10 * 10.).reshape(10, 10, 10))
    a = F(a0)
    assert np.array_equal(a.sreshape((100, 10)).toarray(), a0.reshape((100, 10), order="F"))
    assert np.array_equal(a.sreshape((10, 2, 5, 10)).toarray(), a0.reshape((10, 2, 5, 10), order="F"))
    assert np.array_equal(a.sreshape((10, 1, 10, 1, 10)).toarray(), a0.reshape((10, 1, 10, 1, 10), order="F"))
    assert np.array_equal(a[0:5, :, :].sreshape((5, 10, 5, 2)).toarray(), a0[0:5].reshape((5, 10, 5, 2), order="F"))
    with pytest.raises(ValueError):
        a[0:5, :, :].sreshape((50, 10))  # dims 1,2 of the view are not jointly contiguous
    with pytest.raises(S.DimensionMismatch):
        a.sreshape((7, 7))
    assert a.permutedims((1, 0, 2)).sreshape((10, 10, 5, 2)).strides == (10, 1, 100, 500)


def test_promoteshape_gives_broadcast_dims_stride_zero():
    b1, b2 = F(np.zeros(10)), F(np.zeros((10, 10)))
    p1, p2 = S.promoteshape((10, 10, 10), b1, b2)
    assert p1.strides == (1, 0, 0) and p2.strides == (1, 10, 0) and p1.size == (10, 10, 10)
    with pytest.raises(S.DimensionMismatch):
        S.promoteshape((10, 9), b2)
    one = F(np.zeros((1, 10)))
    assert S.promoteshape1((7, 10), one).strides == (0, 1)


def test_capture_order_is_depth_first_left_to_right():
    a, b, c = F(np.zeros((4, 4))), F(np.zeros((4, 4))), F(np.zeros(4))
    bc = a.adjoint() * b - fn.max(fn.abs(c), fn.real(b))
    leaves = S.capturestridedargs(bc)
    assert [l.strides for l in leaves] == [(4, 1), (1, 4), (1,), (1, 4)]
    cap = S.make_capture(bc)
    assert repr(cap) == "sub(mul(a1, a2), max(abs(a3), real(a4)))"
    code, consts = E.serialize(cap)
    assert list(code) == [OP["ARG"], 1, OP["ARG"], 2, OP["MUL"], 0, OP["ARG"], 3, OP["ABS"], 0, OP["ARG"], 4,
                          OP["REAL"], 0, OP["MAX"], 0, OP["SUB"], 0]
    assert consts == []
    assert S.broadcast_shape(bc) == (4, 4)


def test_nary_plus_folds_left_and_scalars_are_captured():
    x = [E.Arg(i) for i in (1, 2, 3, 4)]
    code, consts = E.serialize(E.Call("add", tuple(x)))
    assert list(code) == [0, 1, 0, 2, OP["ADD"], 0, 0, 3, OP["ADD"], 0, 0, 4, OP["ADD"], 0]
    code, consts = E.serialize(E.trace(lambda a, b: (a + b) / 2, 2))
    assert list(code) == [0, 1, 0, 2, OP["ADD"], 0, OP["CONST"], 0, OP["DIV"], 0] and consts == [2 + 0j]
    code, consts = E.serialize(E.trace(lambda a: a * fn.exp(-2 * a) + fn.sin(a * a), 1))
    assert consts == [-2 + 0j] and len(code) // 2 == 11
    # Ref / Rational / complex constants
    c = S.make_capture(F(np.zeros(3)) * S.Ref(0.5) + Fraction(1, 3) + 2j)
    _, consts = E.serialize(c)
    assert consts[0] == 0.5 and abs(consts[1] - 1 / 3) < 1e-16 and consts[2] == 2j
    with pytest.raises(NotImplementedError):
        E.Call("erf", (E.Arg(1),))


def test_result_eltype_follows_julia_promotion():
    f32, c64 = np.dtype(np.float32), np.dtype(np.complex64)
    t = lambda f, *dts: E.result_dtype(E.trace(f, len(dts)), dts)  # noqa: E731
    assert t(lambda a: a * 2, f32) == f32                  # Int literal does not widen
    assert t(lambda a: a * 2.0, f32) == np.float64         # Float64 literal does
    assert t(lambda a: a * Fraction(1, 3), f32) == f32     # Rational takes the array's type
    assert t(lambda a: a * np.float32(2), f32) == f32
    assert t(lambda a, b: a + b, f32, c64) == c64
    assert t(fn.abs, c64) == f32 and t(fn.real, np.complex128) == np.float64 and t(fn.abs2, c64) == f32
    assert t(lambda a: fn.real(a) < 0, c64) == np.bool_
    assert t(lambda a: a + 1j, f32) == np.complex128
    assert t(fn.sin, np.int32) == np.float64


def test_errors_are_raised_before_the_funnel():
    a, b = F(np.zeros((3, 4))), F(np.zeros((4, 3)))
    with pytest.raises(S.DimensionMismatch):
        S.map_(lambda x: x, a, b)
    with pytest.raises(S.DimensionMismatch):
        S.mapreducedim_(lambda x: x, "+", F(np.zeros((2, 1))), a)
    with pytest.raises(S.DimensionMismatch):
        S.mul_(F(np.zeros((3, 3))), a, a)
    with pytest.raises(ValueError, match="unknown reduction"):
        S.mapreduce(lambda x: x, (lambda p, q: p - q), a)
    assert (a + np.zeros((3, 4))) is not None   # StridedArrayStyle x DefaultArrayStyle: no error (rule 'upload', a5)
    with pytest.raises(TypeError):
        a + [1, 2, 3]                           # not broadcastable at all
    # zero-size map! returns the destination untouched without touching the engine
    z = F(np.zeros((0, 4)))
    assert S.map_(lambda x: x, z, z) is z
    assert S.sum(z) == 0 and S.prod(z) == 1
    with pytest.raises(ValueError):
        S.maximum(z)


def test_initop_tracing_covers_the_five_reference_forms():
    from strided_jl_amd.mapreduce import _initop_code
    from strided_jl_amd import _lib as L
    assert _initop_code(None) == (L.SMR_INIT_NONE, 0j)
    assert _initop_code("identity")[0] == L.SMR_INIT_IDENTITY and _initop_code(lambda x: x)[0] == L.SMR_INIT_IDENTITY
    assert _initop_code("zero")[0] == L.SMR_INIT_ZERO and _initop_code(lambda x: 0)[0] == L.SMR_INIT_ZERO
    assert _initop_code(lambda x: x * (2 - 1j)) == (L.SMR_INIT_SCALE, 2 - 1j)
    assert _initop_code(lambda x: 0.25 * x) == (L.SMR_INIT_SCALE, 0.25 + 0j)
    assert _initop_code(lambda x: 3.5) == (L.SMR_INIT_CONST, 3.5 + 0j)
    assert _initop_code("conj")[0] == L.SMR_INIT_CONJ and _initop_code(fn.conj)[0] == L.SMR_INIT_CONJ
    with pytest.raises(NotImplementedError):
        _initop_code(lambda x: x * x)


def test_traced_f_programs_are_cached_per_closure_including_captured_values():
    """_mapreduce_fuse_ re-traced f on every call.  The cache key holds the values a
    closure captures, so a loop whose scalar changes does not see stale constants."""
    import ctypes as C
    import importlib
    MR = importlib.import_module("strided_jl_amd.mapreduce")  # the package attribute `mapreduce` is the front-end function
    a = S.StridedView(np.zeros((8, 8), order="F"))
    b = a.similar()
    MR._FPROG_CACHE.clear()
    consts = []
    for c in (0.5, 0.75, 0.5):
        p, keep = S.build_problem(lambda x: x * c + 1, None, None, a.size, (b, a), stream=0)
        consts.append([p.fconsts[i] for i in range(2 * p.nconsts)])
    assert consts[0] != consts[1] and consts[0] == consts[2]
    assert len(MR._FPROG_CACHE) == 2          # two distinct captured values, the third call was a hit
    f = lambda x: x * x - 2                   # noqa: E731
    p1, k1 = S.build_problem(f, None, None, a.size, (b, a), stream=0)
    n = len(MR._FPROG_CACHE)
    p2, k2 = S.build_problem(f, None, None, a.size, (b, a), stream=0)
    assert len(MR._FPROG_CACHE) == n and bytes(p1.fprog[0:2 * p1.fprog_len]) == bytes(p2.fprog[0:2 * p2.fprog_len])
    # another operand dtype is another program (ROUND32 placement depends on the types)
    a32 = S.StridedView(np.zeros((8, 8), dtype=np.float32, order="F"))
    S.build_problem(f, None, None, a.size, (b, a32), stream=0)
    assert len(MR._FPROG_CACHE) == n + 1


def test_plain_array_rule_uploads_by_default_and_returns_a_plain_array(monkeypatch):
    """a5 (src/broadcast.jl:11-18, test/othertests.jl:64): Strided x plain Array leaves the strided path in the
    reference and yields a plain Array, never an error.  Default here ('upload'): computed on the views' memory
    space, out-of-place result handed back as a plain host array, in-place destination stays a StridedView;
    strict mode 'error' raises TypeError."""
    import sys

    import oraclelib

    def funnel(f, op, initop, dims, arrays):
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, 1)
        return arrays[0]

    monkeypatch.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
    monkeypatch.setattr(sys.modules["strided_jl_amd.broadcast"], "_mapreduce_fuse_", funnel, raising=False)
    rng = np.random.default_rng(3)
    R1, R2, R3 = rng.random(10), rng.random((10, 10)), rng.random((10, 10, 10))
    B1, B2, B3 = F(R1), F(R2).permutedims((1, 0)), F(R3).permutedims((2, 0, 1))
    A3 = B3.toarray()
    from strided_jl_amd.broadcast import set_plain_array_rule
    old = set_plain_array_rule("error")
    try:
        assert old == "upload"                                 # the default follows the reference: no error
        with pytest.raises(TypeError, match="set_plain_array_rule"):
            B2.adjoint() * A3
    finally:
        set_plain_array_rule(old)
    old = set_plain_array_rule("upload")
    try:
        got = (B2.adjoint() * A3 - fn.max(fn.abs(B1), fn.real(B3))).materialize()
        assert isinstance(got, np.ndarray)                     # "isa Array"
        a1, a2 = B1.toarray(), B2.toarray()
        want = a2.conj().T[:, :, None] * A3 - np.maximum(np.abs(a1)[:, None, None], A3.real)
        assert np.allclose(got, want, rtol=1e-13, atol=0)
        dest = F(np.zeros((10, 10, 10)))
        assert dest.assign(B3 + A3) is dest                    # in place: the destination decides
        assert np.array_equal(dest.toarray(), A3 + A3)
        assert isinstance((B1 + fn.sin(B2 - 3)).materialize(), S.StridedView)   # no plain array: unchanged
    finally:
        set_plain_array_rule(old)
    with pytest.raises(ValueError):
        set_plain_array_rule("cpu")


def test_fprog_cache_never_serves_a_stale_program():
    """the cache keyed captured callables by id() and ignored attributes of global objects.
    A closure is cached only when everything it can read is an immutable scalar."""
    import importlib
    import types
    MR = importlib.import_module("strided_jl_amd.mapreduce")
    a = S.StridedView(np.zeros((8, 8), order="F"))
    b = a.similar()

    def consts_of(f):
        p, keep = S.build_problem(f, None, None, a.size, (b, a), stream=0)
        return [p.fconsts[2 * i] for i in range(p.nconsts)]

    def make(alpha):
        g = lambda x: x * alpha  # noqa: E731
        return lambda x: g(x) + 1

    got = [consts_of(make(al)) for al in (2.0, 3.0, 5.0)]   # fresh inner function per trip, ids may be reused
    assert [c[0] for c in got] == [2.0, 3.0, 5.0]
    cfg = types.SimpleNamespace(alpha=2.0)
    f = lambda x: x * cfg.alpha  # noqa: E731
    assert consts_of(f) == [2.0]
    cfg.alpha = 7.0
    assert consts_of(f) == [7.0]
    assert MR._closure_key(f) is None and MR._closure_key(make(1.0)) is None
    # unhashable / non-scalar defaults: not cached, no TypeError
    h = lambda x, w=[3.0]: x * w[0]  # noqa: E731
    assert MR._closure_key(h) is None
    # module-level scalar read by the closure
    globals()["_SCALE_FOR_TEST"] = 2.0
    k = lambda x: x * _SCALE_FOR_TEST  # noqa: E731,F821
    assert consts_of(k) == [2.0]
    globals()["_SCALE_FOR_TEST"] = 4.0
    assert consts_of(k) == [4.0]
    # the package's own function table is fine (bench / README expressions), nested lambdas are scanned
    fn = S.fn
    assert MR._closure_key(lambda x: x * fn.exp(-2 * x)) is not None
    assert MR._closure_key(lambda x: (lambda y: y * _SCALE_FOR_TEST)(x)) is None  # noqa: F821


def test_numpy_scalar_on_the_left_keeps_its_type():
    """`Float32(0.5) .* A` is a Float32 product in the reference (Julia types every operation of the fused expression); a NumPy scalar
    on the LEFT of a traced argument must not decay to a Python float (= Float64) on its way into the f-program: NumPy does that
    conversion before calling __rmul__ unless the class opts out of the ufunc protocol."""
    import strided_jl_amd.expr as E
    h = np.float32(0.5)
    for e in (h * E.Arg(1), E.Arg(1) * h, h + E.Arg(1), h - E.Arg(1), h / E.Arg(1)):
        consts = [a for a in e.args if isinstance(a, E.Const)]
        assert len(consts) == 1 and consts[0].dtype == np.dtype(np.float32), e
    a = S.StridedView(np.zeros((640, 480, 3), dtype=np.float32, order="F"))
    c = S.StridedView(np.zeros((3, 480, 640), dtype=np.float32, order="F"))
    w = np.float32(2.0)
    d = S.make_plan(lambda x, y: h * x + w * y, None, None, c.size, (c, c, a.permutedims((2, 1, 0)))).describe()
    assert "ct=f32 " in d and "(mixed)" not in d, d
    # ... while a Python float IS a Float64 literal and widens the call, as in Julia
    d = S.make_plan(lambda x, y: 0.5 * x + 2.0 * y, None, None, c.size, (c, c, a.permutedims((2, 1, 0)))).describe()
    assert "ct=f64(mixed)" in d, d


def test_stream_override_is_per_thread():
    """`with S.Stream():` redirects the front ends of the CURRENT thread only (like torch's current stream); no device needed here:
    the stack is exercised directly."""
    import threading
    from strided_jl_amd import mapreduce as M
    import importlib
    M = importlib.import_module("strided_jl_amd.mapreduce")
    seen = {}
    M._push_stream(0x1234)
    try:
        assert M._current_stream() == 0x1234
        t = threading.Thread(target=lambda: seen.setdefault("other", list(M._STREAM_OVERRIDE.stack)))
        t.start()
        t.join()
        assert seen["other"] == []
        M._push_stream(0x5678)
        assert M._current_stream() == 0x5678
        M._pop_stream()
        assert M._current_stream() == 0x1234
    finally:
        M._pop_stream()
    assert M._STREAM_OVERRIDE.stack == []
