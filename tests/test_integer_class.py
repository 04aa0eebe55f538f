"""The integer compute class (round 3; VERDICT r2 missing 4): when every operand has an integer eltype and f stays inside
the integers, the call computes in wrapping 64-bit two's-complement arithmetic like Julia's Int64
(/root/reference/src/mapreduce.jl:55-72: the accumulator type is typeof(op(...)); test/othertests.jl:113-126 reduces
integer-valued views) and truncates on store to a narrower destination.  NumPy's fixed-width integer arithmetic wraps
the same way, so it is the independent truth; the oracle (CPU restatement) and the HIP kernels must agree with it
bit for bit -- above 2^53 and through overflow."""
import numpy as np
import pytest

import strided_jl_amd as S
from util import fview, run_device, run_oracle

fn = S.fn
I64MAX = np.iinfo(np.int64).max


def _big(rng, shape, dtype=np.int64):
    info = np.iinfo(dtype)
    return np.asfortranarray(rng.integers(info.min // 2, info.max // 2, size=shape, dtype=dtype, endpoint=True))


def _cases():
    rng = np.random.default_rng(2026)
    A = _big(rng, (12, 10, 9))
    B = _big(rng, (12, 10, 9))
    S32 = _big(rng, (16, 14), np.int32)
    T32 = _big(rng, (16, 14), np.int32)
    U = np.asfortranarray(rng.integers(0, 2 ** 64 - 1, size=(8, 9), dtype=np.uint64, endpoint=True))
    V = np.asfortranarray(rng.integers(0, 2 ** 64 - 1, size=(8, 9), dtype=np.uint64, endpoint=True))
    small = np.asfortranarray(rng.integers(-100, 100, size=(7, 6, 5), dtype=np.int64))
    with np.errstate(over="ignore"):
        cases = {
            # name: (f, op, initop, dims, [dest array, inputs as (array, perm or None)], expected)
            "add_above_2^53": (lambda a, b: a + b, None, None, A.shape, [np.zeros_like(A), (A, None), (B, None)], A + B),
            "mul_wraps": (lambda a, b: a * b, None, None, A.shape, [np.zeros_like(A), (A, None), (B, None)], A * B),
            "expr_permuted": (lambda a, b: a * 3 - b + 7, None, None, (9, 10, 12),
                              [np.zeros((9, 10, 12), dtype=np.int64, order="F"), (A, (2, 1, 0)), (B, (2, 1, 0))],
                              A.transpose(2, 1, 0) * 3 - B.transpose(2, 1, 0) + 7),
            "neg_abs_min_max": (lambda a, b: fn.max(fn.abs(a), -b) - fn.min(a, b), None, None, A.shape,
                                [np.zeros_like(A), (A, None), (B, None)], np.maximum(np.abs(A), -B) - np.minimum(A, B)),
            "abs2": (fn.abs2, None, None, A.shape, [np.zeros_like(A), (A, None)], A * A),
            "compare_to_bool": (lambda a, b: a < b, None, None, A.shape, [np.zeros(A.shape, dtype=np.uint8, order="F"), (A, None), (B, None)],
                                (A < B).astype(np.uint8)),
            "int32_wraps_in_int32": (lambda a, b: a * b + a, None, None, S32.shape, [np.zeros_like(S32), (S32, None), (T32, None)], S32 * T32 + S32),
            "int32_times_int64": (lambda a, b: a * b, None, None, S32.shape,
                                  [np.zeros(S32.shape, dtype=np.int64, order="F"), (S32, None), (S32.astype(np.int64) << 20, None)],
                                  S32.astype(np.int64) * (S32.astype(np.int64) << 20)),
            "uint64_ring": (lambda a, b: a * b + a - b, None, None, U.shape, [np.zeros_like(U), (U, None), (V, None)], U * V + U - V),
            "sum_all_above_2^53": (lambda a: a, "+", None, A.shape, [np.zeros((1, 1, 1), dtype=np.int64), (A, None)], np.array(A.sum(dtype=np.int64)).reshape(1, 1, 1)),
            "sum_dims_02_permuted": (lambda a: a, "+", None, (9, 10, 12), [np.zeros((1, 10, 1), dtype=np.int64, order="F"), (A, (2, 1, 0))],
                                     A.transpose(2, 1, 0).sum(axis=(0, 2), keepdims=True, dtype=np.int64)),
            "prod_wraps": (lambda a: a, "*", None, small.shape, [np.ones((7, 1, 1), dtype=np.int64, order="F"), (small, None)],
                           np.multiply.reduce(np.multiply.reduce(small, axis=2, keepdims=True), axis=1, keepdims=True)),
            "min_dims": (lambda a: a, "min", None, A.shape, [np.full((12, 1, 9), I64MAX, dtype=np.int64, order="F"), (A, None)], A.min(axis=1, keepdims=True)),
            "max_of_abs2_diff": (lambda a, b: fn.abs2(a - b), "max", None, small.shape,
                                 [np.full((1, 6, 1), np.iinfo(np.int64).min, dtype=np.int64, order="F"), (small, None), (small[::-1, :, :].copy(order="F"), None)],
                                 ((small - small[::-1, :, :]) ** 2).max(axis=(0, 2), keepdims=True)),
            "sum_int32_into_int64": (lambda a: a, "+", None, S32.shape, [np.zeros((1, 1), dtype=np.int64), (S32, None)], np.array(S32.sum(dtype=np.int64)).reshape(1, 1)),
            "sum_int32_into_int32_wraps": (lambda a: a, "+", None, S32.shape, [np.zeros((16, 1), dtype=np.int32, order="F"), (S32, None)],
                                           S32.sum(axis=1, keepdims=True, dtype=np.int32)),
            "accumulate_with_scale_initop": (lambda a: a, "+", ("scale", 3), small.shape, [np.full((7, 6, 1), 5, dtype=np.int64, order="F"), (small, None)],
                                             5 * 3 + small.sum(axis=2, keepdims=True)),
        }
    return cases


CASES = _cases()


def _views(case):
    f, op, initop, dims, arrs, want = case
    dest = fview(arrs[0])
    ins = []
    for a, perm in arrs[1:]:
        v = fview(a)
        ins.append(v.permutedims(perm) if perm is not None else v)
    # reduced dims of the destination: stride 0 over the full box
    if op is not None:
        st = tuple(0 if n == 1 and d != 1 else s for n, s, d in zip(dest.size, dest.strides, dims))
        dest_full = S.StridedView(dest.parent, tuple(dims), st, dest.offset)
    else:
        dest_full = dest
    return f, op, initop, tuple(dims), (dest_full,) + tuple(ins), dest, want


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("nthreads", [1, 3])
def test_oracle_integer_class_equals_numpy(name, nthreads):
    f, op, initop, dims, arrays, dest, want = _views(CASES[name])
    run_oracle(f, op, initop, dims, arrays, nthreads)
    got = dest.toarray()
    assert got.dtype == want.dtype and np.array_equal(got, want.reshape(got.shape)), name


def test_planner_selects_the_integer_class_and_compiles_its_functors():
    rng = np.random.default_rng(1)
    A = fview(_big(rng, (64, 48)))
    plan = S.make_plan(lambda a, b: a * b - 3, None, None, A.size, (A.similar(), A, A))
    assert " ct=i64" in plan.describe()
    assert plan.jit_compile() > 0            # hiprtc, gfx950, integer functor (cross-compiles without a GPU)
    plan = S.make_plan(lambda a: a, "+", None, A.size, (S.StridedView(np.zeros(1, dtype=np.int64), A.size, (0, 0), 0), A))
    assert " ct=i64" in plan.describe() and "reduce_all" in plan.describe()
    # a pure move stays a bit copy; integers that meet floating-point arithmetic leave the class
    assert "(bitcopy)" in S.make_plan(lambda a: a, None, None, A.size, (A.similar(), A.permutedims((1, 0)).permutedims((1, 0)))).describe()
    F = fview(rng.random((64, 48)))
    assert " ct=f64" in S.make_plan(lambda a, b: a * b, None, None, A.size, (F.similar(), F, fview(_big(rng, (64, 48), np.int32)))).describe()


def test_what_the_integer_class_refuses():
    rng = np.random.default_rng(2)
    A = fview(_big(rng, (8, 8)))
    U = fview(np.asfortranarray(rng.integers(0, 2 ** 63, size=(8, 8), dtype=np.uint64)))
    with pytest.raises(S._lib.UnsupportedOnDevice):     # Int64 input meets a division: Float64 would round above 2^53
        S.make_plan(lambda a: a / 2, None, None, A.size, (A.similar(np.float64), A))
    with pytest.raises(S._lib.UnsupportedOnDevice):     # no order on UInt64 in the 64-bit signed class
        S.make_plan(lambda a, b: fn.min(a, b), None, None, U.size, (U.similar(), U, U))
    with pytest.raises(S._lib.UnsupportedOnDevice):
        S.make_plan(lambda a: a, "max", None, U.size, (S.StridedView(np.zeros(1, dtype=np.uint64), U.size, (0, 0), 0), U))
    # ring operations on UInt64 are fine
    assert " ct=i64" in S.make_plan(lambda a, b: a * b + a, None, None, U.size, (U.similar(), U, U)).describe()


def test_front_ends_on_integer_views_use_the_integer_class(monkeypatch):
    """S.sum / S.maximum / map on Int64 views (NumPy's default integer): values above 2^53 stay exact."""
    import sys

    import oraclelib

    def funnel(f, op, initop, dims, arrays):
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, 1)
        return arrays[0]

    monkeypatch.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
    monkeypatch.setattr(sys.modules["strided_jl_amd.broadcast"], "_mapreduce_fuse_", funnel, raising=False)
    rng = np.random.default_rng(3)
    a = rng.integers(2 ** 60, 2 ** 61, size=(5, 7), dtype=np.int64)
    A = fview(a)
    assert S.sum(A) == int(a.sum()) and S.maximum(A) == int(a.max()) and S.minimum(A) == int(a.min())
    assert np.array_equal(S.sum(A, dims=0).toarray(), a.sum(axis=0, keepdims=True))
    assert np.array_equal(S.map(lambda x, y: x * 2 - y, A, A).toarray(), a * 2 - a)
    assert np.array_equal((A + 1).materialize().toarray(), a + 1)
    s32 = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(9, 4), dtype=np.int32)
    assert S.sum(fview(s32)) == int(s32.sum(dtype=np.int64))     # Base.add_sum widens small integers


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_integer_class_equals_oracle_and_numpy(name):
    f, op, initop, dims, arrays, dest, want = _views(CASES[name])
    got_dev = run_device(f, op, initop, dims, arrays)
    run_oracle(f, op, initop, dims, arrays, 1)
    got_dev = np.asarray(got_dev)
    # run_device returns the destination over the full box for reductions (stride-0 dims): compare the kept elements
    kept = tuple(slice(0, 1) if (s == 0 and n > 1) else slice(None) for n, s in zip(arrays[0].size, arrays[0].strides))
    got_dev = got_dev[kept]
    ora = dest.toarray()
    assert np.array_equal(got_dev.reshape(want.shape), want), name + ": HIP vs NumPy"
    assert np.array_equal(got_dev.reshape(ora.shape), ora), name + ": HIP vs oracle"


@pytest.mark.gpu
@pytest.mark.parametrize("shape,perm", [((64, 64, 16), (2, 0, 1)), ((4000, 30), (1, 0)), ((32, 32, 32, 8), (3, 2, 1, 0))])
def test_hip_integer_maps_across_kernel_families(shape, perm):
    """Transposing (TILED), streaming and generic families in the integer class, Int64 and Int32 data, JIT functors."""
    import torch
    rng = np.random.default_rng(5)
    for dt in (np.int64, np.int32):
        a = _big(rng, shape, dt)
        b = _big(rng, tuple(shape[i] for i in perm), dt)

        def dv(x):
            t = torch.from_numpy(np.asfortranarray(x).ravel(order="F").copy()).cuda()
            st, s = [], 1
            for d in x.shape:
                st.append(s)
                s *= d
            return S.StridedView(t, x.shape, tuple(st), 0)

        A, B = dv(a), dv(b)
        out = B.similar()
        S.map_(lambda x, y: x * y - (x + 5), out, A.permutedims(perm), B)
        with np.errstate(over="ignore"):
            want = a.transpose(perm) * b - (a.transpose(perm) + dt(5))
        assert np.array_equal(out.toarray(), want), (shape, perm, dt)
        tot = S.sum(A)
        assert tot == int(a.sum(dtype=np.int64))
        with np.errstate(over="ignore"):
            assert np.array_equal(S.sum(A, dims=0).toarray(), a.sum(axis=0, keepdims=True, dtype=np.int64))
