"""The parity case list: the reference's own test-suite (/root/reference/test/othertests.jl)
re-expressed over the Python mirror of its interface.  Every case is a function of `mk`
(ndarray -> StridedView over a fresh column-major copy, on the host or on the device) that
returns (results, expected, exact): results are produced through the public front-ends
(map_, broadcast, mapreduce, mul_ ...), `expected` is the Base-Julia side of the reference's
assertion restated with NumPy, and `exact` says whether the reference compares with `==`
(bit-exact) or `isapprox` (rtol = sqrt(eps)).

The same cases run (a) with the funnel patched to the CPU oracle, compared with NumPy
(tests/test_oracle_numpy.py, no GPU) and (b) on the HIP path, compared with the oracle and with
NumPy (tests/test_gpu_parity.py).
"""
from fractions import Fraction

import numpy as np

import strided_jl_amd as S
from strided_jl_amd import fn

FLOATS = (np.float32, np.float64, np.complex64, np.complex128)


def _rand(rng, shape, T):
    T = np.dtype(T)
    if np.issubdtype(T, np.complexfloating):
        x = rng.random(shape) + 1j * rng.random(shape)
    else:
        x = rng.random(shape)
    return np.asfortranarray(x.astype(T))


def _randn(rng, shape, T):
    T = np.dtype(T)
    if np.issubdtype(T, np.complexfloating):
        x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    else:
        x = rng.standard_normal(shape)
    return np.asfortranarray(x.astype(T))


def _randperm(rng, n):
    return tuple(int(i) for i in rng.permutation(n))


CASES = []


def case(name, **params):
    def deco(fn_):
        CASES.append((name, fn_, params))
        return fn_
    return deco


def _register_typed(name, T):
    def deco(fn_):
        CASES.append((f"{name}[{np.dtype(T).name}]", lambda mk, rng, fn_=fn_, T=T: fn_(mk, rng, T), {}))
        return fn_
    return deco


def typed(name):
    def deco(fn_):
        for T in FLOATS:
            CASES.append((f"{name}[{np.dtype(T).name}]", (lambda mk, rng, fn_=fn_, T=T: fn_(mk, rng, T)), {}))
        return fn_
    return deco


# ---- test/othertests.jl:1-15  in-place matrix operations (==) ---------------------------------------
@typed("inplace_matrix_ops")
def inplace_matrix_ops(mk, rng, T, n=(257, 130)):
    A1 = _randn(rng, n, T)
    A2t = np.zeros((n[1], n[0]), dtype=T, order="F")
    res, exp = [], []
    B1 = mk(A1)
    res.append(S.conj_(B1).toarray()); exp.append(np.conj(A1))
    A1 = np.conj(A1)  # conj! mutated both sides
    B1 = mk(A1)
    B2 = mk(A2t)
    res.append(S.adjoint_(B2, B1).toarray()); exp.append(np.conj(A1.T))
    res.append(S.transpose_(B2, B1).toarray()); exp.append(A1.T)
    res.append(S.permutedims_(B2, B1, (1, 0)).toarray()); exp.append(A1.T)
    return res, exp, True


@case("inplace_matrix_ops_1000_f64")  # the reference's size: 1e6 elements > MINTHREADLENGTH
def inplace_1000(mk, rng):
    return inplace_matrix_ops(mk, rng, np.float64, n=(1000, 1000))


@case("inplace_matrix_ops_1000_c64")
def inplace_1000c(mk, rng):
    return inplace_matrix_ops(mk, rng, np.complex64, n=(1000, 1000))


# ---- :17-44  map, scale!, axpy!, axpby! --------------------------------------------------------------
def _blas1(mk, rng, T, N):
    dims = (60 // N,) * N
    R1, R2, R3 = _rand(rng, dims, T), _rand(rng, dims, T), _rand(rng, dims, T)
    p1, p2, p3 = _randperm(rng, N), _randperm(rng, N), _randperm(rng, N)
    B1, B2, B3 = mk(R1).permutedims(p1), mk(R2).permutedims(p2), mk(R3).permutedims(p3)
    A1, A2, A3 = R1.transpose(p1).copy(), R2.transpose(p2).copy(), R3.transpose(p3).copy()
    res, exp = [], []
    res.append(S.Array(B1)); exp.append(A1)  # convert(Array, B1)
    half, third = Fraction(1, 2), Fraction(1, 3)

    def c(fr):  # Float32 * Rational -> Float32 arithmetic in Julia; a weak Python float in NumPy 2
        return fr.numerator / fr.denominator

    A1 = (A1 * c(half)).astype(T); res.append(S.rmul_(B1, half).toarray()); exp.append(A1.copy())
    A2 = (c(third) * A2).astype(T); res.append(S.lmul_(third, B2).toarray()); exp.append(A2.copy())
    A2 = (c(third) * A1 + A2).astype(T); res.append(S.axpy_(third, B1, B2).toarray()); exp.append(A2.copy())
    A3 = (A2 + A3).astype(T); res.append(S.axpy_(1, B2, B3).toarray()); exp.append(A3.copy())
    A3 = (c(third) * A1 + c(half) * A3).astype(T); res.append(S.axpby_(third, B1, half, B3).toarray()); exp.append(A3.copy())
    A1 = (A2 + A1).astype(T); res.append(S.axpby_(1, B2, 1, B1).toarray()); exp.append(A1.copy())
    f = lambda x, y, z: fn.sin(x) + y / fn.exp(-fn.abs(z))  # noqa: E731
    m = S.map(f, B1, B2, B3)
    assert isinstance(m, S.StridedView)
    res.append(m.toarray()); exp.append(np.sin(A1) + A2 / np.exp(-np.abs(A3)))
    res.append(S.mul_(B1, 1, B2).toarray()); exp.append(A2.copy())
    res.append(S.mul_(B1, B2, 1).toarray()); exp.append(A2.copy())
    return res, exp, False


for _T in FLOATS:
    for _N in range(2, 7):
        CASES.append((f"map_scale_axpy_axpby[{np.dtype(_T).name}-N{_N}]",
                      (lambda mk, rng, T=_T, N=_N: _blas1(mk, rng, T, N)), {}))


# ---- :46-66  broadcast with mixed ranks / Ref -------------------------------------------------------
@typed("broadcast")
def broadcast(mk, rng, T):
    R1, R2, R3 = _rand(rng, (10,), T), _rand(rng, (10, 10), T), _rand(rng, (10, 10, 10), T)
    p2, p3 = _randperm(rng, 2), _randperm(rng, 3)
    B1, B2, B3 = mk(R1), mk(R2).permutedims(p2), mk(R3).permutedims(p3)
    A1, A2, A3 = R1, R2.transpose(p2), R3.transpose(p3)
    a1 = A1.reshape(10, 1, 1)
    res, exp = [], []
    r = S.materialize(B1 + fn.sin(B2 - 3))
    assert isinstance(r, S.StridedView)
    res.append(r.toarray()); exp.append(A1.reshape(10, 1) + np.sin(A2 - 3))
    r = S.materialize(B2.adjoint() * B3 - S.Ref(0.5))
    res.append(r.toarray()); exp.append(np.conj(A2.T)[:, :, None] * A3 - 0.5)
    r = S.materialize(B2.adjoint() * B3 - fn.max(fn.abs(B1), fn.real(B3)))
    res.append(r.toarray()); exp.append(np.conj(A2.T)[:, :, None] * A3 - np.maximum(np.abs(a1), np.real(A3)))
    return res, exp, False


# ---- :68-107  partial reductions, every initop form ---------------------------------------------------
@typed("mapreduce_partial")
def mapreduce_partial(mk, rng, T):
    R1 = _rand(rng, (10,) * 6, T)
    res, exp = [], []
    res.append(S.sum(mk(R1), dims=(0, 2, 4)).toarray()); exp.append(R1.sum(axis=(0, 2, 4), keepdims=True))
    res.append(S.mapreduce(fn.sin, "+", mk(R1), dims=(0, 2, 4)).toarray())
    exp.append(np.sin(R1).sum(axis=(0, 2, 4), keepdims=True))
    R2 = _rand(rng, (10, 10, 10), T)
    red = np.sin(R1).sum(axis=(1, 2, 5), keepdims=True)
    r2 = R2.reshape((10, 1, 1, 10, 10, 1), order="F")
    beta = (rng.random() + (1j * rng.random() if np.issubdtype(T, np.complexfloating) else 0))
    beta = np.dtype(T).type(beta).item()
    forms = [
        ("identity", red + r2),
        ((lambda x: 0), red),
        ((lambda x: beta * x), red + beta * r2),
        ((lambda x: beta), red + beta),
        ("conj", red + np.conj(r2)),
    ]
    dims = (10,) * 6
    for initop, expected in forms:
        out = mk(R2).sreshape((10, 1, 1, 10, 10, 1))
        S._mapreducedim_(fn.sin, "+", initop, dims, (out, mk(R1)))
        res.append(out.toarray()); exp.append(expected)
    R3 = _rand(rng, (100, 100, 2), T)
    res.append(S.sum(mk(R3), dims=(0, 1)).toarray()); exp.append(R3.sum(axis=(0, 1), keepdims=True))
    return res, exp, False


# ---- :109-128  complete reductions --------------------------------------------------------------------
@typed("mapreduce_complete")
def mapreduce_complete(mk, rng, T):
    R1 = _rand(rng, (10,) * 6, T)
    res, exp = [], []

    RT = np.finfo(T).dtype  # real type of T

    def all_(V, A):
        wide = np.complex128 if np.iscomplexobj(A) else np.float64
        res.append(np.asarray(S.sum(V), dtype=T)); exp.append(np.asarray(A.sum(dtype=wide), dtype=T))
        res.append(np.asarray(S.maximum(V, f=fn.abs), dtype=RT)); exp.append(np.asarray(np.abs(A).max(), dtype=RT))
        res.append(np.asarray(S.minimum(V, f=fn.real), dtype=RT)); exp.append(np.asarray(np.real(A).min(), dtype=RT))

    all_(mk(R1), R1)
    p = _randperm(rng, 6)
    all_(mk(R1).permutedims(p), R1.transpose(p))
    R3 = _rand(rng, (5, 5, 5), T)
    wide = np.complex128 if np.iscomplexobj(R3) else np.float64
    res.append(np.asarray(S.prod(mk(R3), f=fn.exp), dtype=T)); exp.append(np.asarray(np.exp(R3.sum(dtype=wide)), dtype=T))
    return res, exp, False


@typed("count_negative_real")  # sum(x -> real(x) < 0, A) compared with == (:116,:123)
def count_negative(mk, rng, T):
    R1 = _randn(rng, (10,) * 6, T)
    p = _randperm(rng, 6)
    cnt = int((np.real(R1) < 0).sum())
    res = [np.asarray(S.sum(mk(R1), f=lambda x: fn.real(x) < 0)),
           np.asarray(S.sum(mk(R1).permutedims(p), f=lambda x: fn.real(x) < 0))]
    return res, [np.asarray(cnt), np.asarray(cnt)], True


# ---- :130-190  @strided over views / reshapes -----------------------------------------------------------
@typed("strided_views")
def strided_views(mk, rng, T):
    A1, A2, A3 = _rand(rng, (10,), T), _rand(rng, (10, 10), T), _rand(rng, (10, 10, 10), T)
    V1, V2, V3 = mk(A1), mk(A2), mk(A3)
    res, exp = [], []
    # view(A2, :, 1:2:10)
    r = S.materialize(V1 + fn.sin(V2[:, 0:10:2] - 3))
    res.append(r.toarray()); exp.append(A1[:, None] + np.sin(A2[:, 0:10:2] - 3))
    # view(A2', :, 1:6) .* view(A3, :, 1:6, 4) .- Ref(0.5)
    r = S.materialize(V2.adjoint()[:, 0:6] * V3[:, 0:6, 3] - S.Ref(0.5))
    res.append(r.toarray()); exp.append(np.conj(A2.T)[:, 0:6] * A3[:, 0:6, 3] - 0.5)
    # view(A2,:,3)' .* view(A3,1:5,:,2:2:10) .- max.(abs.(view(A1,1:5)), real.(view(A3,4:4,4:4,2:2:10)))
    B2 = V2[:, 2:3].adjoint()           # (1, 10): adjoint of a column vector
    B3 = V3[0:5, :, 1:10:2]
    B1 = V1[0:5]
    B3b = V3[3:4, 3:4, 1:10:2]
    r = S.materialize(B2 * B3 - fn.max(fn.abs(B1), fn.real(B3b)))
    e = np.conj(A2[:, 2])[None, :, None] * A3[0:5, :, 1:10:2] - np.maximum(
        np.abs(A1[0:5])[:, None, None], np.real(A3[3:4, 3:4, 1:10:2]))
    res.append(r.toarray()); exp.append(e)
    # reshape(A2, (10, 2, 5))
    r = S.materialize(V1 + fn.sin(V2.sreshape((10, 2, 5)) - 3))
    res.append(r.toarray()); exp.append(A1[:, None, None] + np.sin(A2.reshape((10, 2, 5), order="F") - 3))
    # reshape(A2, 1, 100)' .* reshape(A3, 100, 1, 10) .- Ref(0.5)
    r = S.materialize(V2.sreshape((1, 100)).adjoint() * V3.sreshape((100, 1, 10)) - S.Ref(0.5))
    e = np.conj(A2.reshape((1, 100), order="F").T)[:, :, None] * A3.reshape((100, 1, 10), order="F") - 0.5
    res.append(r.toarray()); exp.append(e)
    # reshape(view(A3, 1:5, :, :), 5, 10, 5, 2)
    B3 = V3[0:5, :, :].sreshape((5, 10, 5, 2))
    r = S.materialize(B2.sreshape((1, 10)) * B3 - fn.max(fn.abs(B1), fn.real(B3b.sreshape((1, 1, 5)))))
    e = np.conj(A2[:, 2])[None, :, None, None] * A3[0:5].reshape((5, 10, 5, 2), order="F") - np.maximum(
        np.abs(A1[0:5])[:, None, None, None], np.real(A3[3:4, 3:4, 1:10:2]).reshape((1, 1, 5, 1), order="F"))
    res.append(r.toarray()); exp.append(e)
    return res, exp, False


# ---- beyond the reference's tests: strides it never exercises (SURVEY section 4 "not tested") -------
@typed("negative_strides_and_offsets")
def negative_strides(mk, rng, T):
    A = _rand(rng, (12, 9, 7), T)
    V = mk(A)
    res, exp = [], []
    res.append(S.copy(V[::-1, :, ::-2]).toarray()); exp.append(A[::-1, :, ::-2])
    D = mk(np.zeros((12, 9, 7), dtype=T))
    D[::-1, ::-1, :] = V + V
    res.append(D.toarray()); exp.append((A + A)[::-1, ::-1, :])
    res.append(S.Array(V[2:11:3, 1:, 6].transpose())); exp.append(A[2:11:3, 1:, 6].T)
    return res, exp, True


@case("integer_permute_is_a_bit_move")
def integer_permute(mk, rng):
    res, exp = [], []
    for T in (np.int8, np.int16, np.int32, np.int64, np.uint8):
        A = np.asfortranarray(rng.integers(np.iinfo(T).min, np.iinfo(T).max, size=(9, 8, 7, 6), dtype=T))
        p = _randperm(rng, 4)
        B = mk(np.zeros(tuple(A.shape[i] for i in p), dtype=T))
        res.append(S.permutedims_(B, mk(A), p).toarray()); exp.append(A.transpose(p))
    return res, exp, True


@case("nan_payload_and_signed_zero_survive_copy")
def nan_payload(mk, rng):
    A = np.asfortranarray(rng.standard_normal((33, 17)))
    bits = A.view(np.uint64)
    bits[3, 4] = 0x7FF8000000000ABC  # quiet NaN with payload
    bits[5, 6] = 0x7FF0000000000123 | (1 << 51)
    bits[7, 8] = 0x8000000000000000  # -0.0
    B = mk(np.zeros((17, 33)))
    r = S.permutedims_(B, mk(A), (1, 0)).toarray()
    return [r.view(np.uint64)], [np.ascontiguousarray(A.T).view(np.uint64)], True


# ---- :253-333  generic matmul (3-operand reduce + initop); integer-valued so results are exact --------
def _opn(a, op):
    return {"identity": a, "conj": np.conj(a), "transpose": a.T, "adjoint": np.conj(a.T)}[op]


def _opv(v, op):
    return {"identity": v, "conj": v.conj(), "transpose": v.transpose(), "adjoint": v.adjoint()}[op]


def _matmul(mk, rng, d, ops1, ops2, ops3):
    ri = lambda: np.asfortranarray(rng.integers(-100, 101, (d, d)) + 1j * rng.integers(-100, 101, (d, d)))  # noqa: E731
    A1, A2, A4 = ri(), ri(), ri()
    alpha, beta = 2 + 1j, 3 - 1j
    res, exp = [], []
    for o1 in ops1:
        for o2 in ops2:
            P = _opn(A1, o1) @ _opn(A2, o2)
            for o3 in ops3:
                cb = np.conj(beta) if o3 in ("conj", "adjoint") else beta
                for (a, b, e) in ((alpha, beta, cb * A4 + _opn(alpha * P, o3)),
                                  (alpha, 0, _opn(alpha * P, o3)),
                                  (1, 0, _opn(P, o3)),
                                  (1, 1, A4 + _opn(P, o3))):
                    B3 = mk(A4)
                    S.mul_(_opv(B3, o3), _opv(mk(A1), o1), _opv(mk(A2), o2), a, b)
                    res.append(B3.toarray()); exp.append(e)
    return res, exp, True


ALLOPS = ("identity", "conj", "transpose", "adjoint")


@case("generic_matmul_all_op_combinations_d13")
def matmul_small(mk, rng):
    return _matmul(mk, rng, 13, ALLOPS, ALLOPS, ALLOPS)


@case("generic_matmul_d103")
def matmul_103(mk, rng):
    return _matmul(mk, rng, 103, ("identity", "adjoint"), ("transpose",), ("identity", "conj"))


@case("zero_size_rules")
def zero_size(mk, rng):
    # map! over an empty box returns the destination untouched (src/mapreduce.jl:48);
    # _mapreducedim! with a zero-size reduction dim still applies initop (:88-91)
    res, exp = [], []
    C0 = np.asfortranarray(rng.integers(-100, 101, (2, 2)) + 1j * rng.integers(-100, 101, (2, 2)))
    A = np.zeros((2, 0), dtype=np.complex128, order="F")
    B = np.zeros((0, 2), dtype=np.complex128, order="F")
    C = mk(C0)
    S.mul_(C, mk(A), mk(B), 3 + 2j, 1)
    res.append(C.toarray()); exp.append(C0)
    C = mk(C0)
    S.mul_(C, mk(A), mk(B), 3 + 2j, 2)
    res.append(C.toarray()); exp.append(2 * C0)
    out = mk(C0).sreshape((2, 2, 1))
    S._mapreducedim_(lambda x: x, "+", "zero", (2, 2, 0), (out, mk(np.zeros((2, 2, 0), dtype=np.complex128, order="F"))))
    res.append(out.toarray()); exp.append(np.zeros((2, 2, 1)))
    return res, exp, True


# ---- the README / BASELINE workloads at small size ---------------------------------------------------
@case("readme_workloads_small")
def readme_small(mk, rng):
    res, exp = [], []
    A = _randn(rng, (96, 96), np.float64)
    B = mk(np.zeros_like(A))
    V = mk(A)
    B.assign((V + V.adjoint()) / 2)
    res.append(B.toarray()); exp.append((A + A.T) / 2)
    B.assign(3 * V.adjoint())
    res.append(B.toarray()); exp.append(3 * A.T)
    A4 = _randn(rng, (12, 10, 9, 11), np.float64)
    V4 = mk(A4)
    B4 = mk(np.zeros((11, 9, 10, 12)))
    S.permutedims_(B4, V4, (3, 2, 1, 0))
    res.append(B4.toarray()); exp.append(A4.transpose(3, 2, 1, 0))
    C = _randn(rng, (8, 8, 8, 8), np.float64)
    VC = mk(C)
    D = mk(np.zeros_like(C))
    D.assign(VC.permutedims((0, 1, 2, 3)) + VC.permutedims((1, 2, 3, 0)) + VC.permutedims((2, 3, 0, 1)) +
             VC.permutedims((3, 0, 1, 2)))
    res.append(D.toarray())
    exp.append(((C + C.transpose(1, 2, 3, 0)) + C.transpose(2, 3, 0, 1)) + C.transpose(3, 0, 1, 2))
    return res, exp, True


def run_case(name, mk, seed=1234):
    for n, f, _ in CASES:
        if n == name:
            return f(mk, np.random.default_rng(seed))
    raise KeyError(name)


NAMES = [n for n, _, _ in CASES]
