"""GPU: the kernel-side changes of round 4, each against NumPy (bit-exact where the reference compares with `==`):
  * stepped ranges behind a permutation (an input whose smallest stride is 2..4 elements) take the TILED family with that dim as the
    input's near-unit axis (csrc/smr_plan.cpp: near_axis) -- /root/reference/test/othertests.jl:130-190 uses exactly such views;
  * STREAM rows of 129 .. 1024 elements share a workgroup (packed form of csrc/smr_k_stream.hip);
  * split partial reductions with up to 1024 chunks fold inside the launch through two levels of arrival counters
    (csrc/smr_k_reduce.hip: fold_in_launch) -- /root/reference/test/othertests.jl:68-107."""
import numpy as np
import pytest

import strided_jl_amd as S

pytestmark = pytest.mark.gpu


def dview(arr):
    import torch
    a = np.asfortranarray(arr)
    t = torch.from_numpy(a.ravel(order="F").copy()).cuda()
    st, s = [], 1
    for d in a.shape:
        st.append(s)
        s *= d
    return S.StridedView(t, a.shape, tuple(st), 0)


def sync():
    import torch
    torch.cuda.synchronize()


@pytest.mark.parametrize("step", [2, 3, 4])
@pytest.mark.parametrize("dt", [np.float64, np.float32, np.complex64])
def test_stepped_range_behind_a_transpose(step, dt):
    rng = np.random.default_rng(step)
    n = 2048
    m = (n - 2) // step
    a = rng.standard_normal((n, n)).astype(dt)
    if np.issubdtype(dt, np.complexfloating):
        a = (a + 1j * rng.standard_normal((n, n))).astype(dt)
    A, B = dview(a), dview(np.zeros((n, n), dtype=dt))
    dest = B.sview(slice(0, m), slice(0, m))
    x = A.sview(slice(0, m * step, step), slice(0, m))                              # strides (step, n)
    y = A.permutedims((1, 0)).sview(slice(0, m), slice(0, m * step, step))            # strides (n, step): no unit axis at all
    plan = S.make_plan(lambda u, v: u + v, None, None, (m, m), (dest, x, y))
    assert "family=tiled" in plan.describe(), plan.describe()
    plan.execute()
    sync()
    want = a[0:m * step:step, 0:m] + a.T[0:m, 0:m * step:step]
    got = B.toarray()
    assert np.array_equal(got[:m, :m], want)
    assert not got[m:, :].any() and not got[:, m:].any()                             # nothing outside the destination view


def test_stepped_range_with_offsets_conj_and_three_dims():
    rng = np.random.default_rng(11)
    a = (rng.standard_normal((90, 70, 66)) + 1j * rng.standard_normal((90, 70, 66))).astype(np.complex128)
    A = dview(a)
    v = A.sview(slice(3, 87, 3), slice(1, 69, 2), slice(2, 64)).permutedims((2, 0, 1))  # (62, 28, 34), strides (6300, 3, 180)
    out = dview(np.zeros(v.size, dtype=np.complex128))
    S.map_(lambda z: S.fn.conj(z) * 2, out, v)
    sync()
    want = np.conj(np.transpose(a[3:87:3, 1:69:2, 2:64], (2, 0, 1))) * 2
    assert np.array_equal(out.toarray(), want)


@pytest.mark.parametrize("shape,perm", [((257, 129, 65), (0, 2, 1)), ((513, 40, 9), (0, 2, 1)), ((1001, 33, 7), (0, 2, 1)), ((130, 300), (0, 1)),
                                        ((999, 64, 3, 5), (0, 3, 1, 2))])
@pytest.mark.parametrize("dt", [np.float64, np.float32, np.complex128])
def test_packed_stream_rows(shape, perm, dt):
    rng = np.random.default_rng(len(shape))
    a = rng.standard_normal(shape).astype(dt)
    A = dview(a)
    out = dview(np.zeros(tuple(shape[i] for i in perm), dtype=dt))
    res = {}
    for pack in (1, 0):
        S.set_option("stream_pack_rows", pack)
        try:
            plan = S.make_plan(lambda x: x * 3 - 1, None, None, out.size, (out, A.permutedims(perm)))
            if len(shape) > 2:
                assert "family=stream" in plan.describe(), plan.describe()
            out.parent.zero_()
            plan.execute()
            sync()
            res[pack] = out.toarray()
        finally:
            S.set_option("stream_pack_rows", 1)
    want = np.transpose(a, perm) * dt(3) - dt(1)
    assert np.array_equal(res[1], want) and np.array_equal(res[0], want)


RED_SHAPES = [((100, 90, 80, 7), (1, 2, 3), np.float32), ((100, 90, 80, 7), (1, 2), np.float32), ((100, 90, 80, 7), (0, 2, 3), np.float32),
              ((100, 90, 80, 7), (0, 1, 2), np.float64), ((512, 384, 64), (1, 2), np.float32), ((512, 384, 64), (0, 1), np.float64),
              ((33, 100000), (1,), np.float64), ((7, 3, 250000), (2,), np.float32)]


@pytest.mark.parametrize("shape,dims,dt", RED_SHAPES)
def test_two_level_in_launch_fold(shape, dims, dt):
    rng = np.random.default_rng(sum(shape))
    a = rng.integers(-8, 9, size=shape).astype(dt)          # small integers: every summation order gives the same float
    A = dview(a)
    got = {}
    for tree in (1024, 0):
        S.set_option("reduce_tree", tree)
        try:
            r1 = S.sum(A, dims=dims).toarray()
            r2 = S.sum(A, dims=dims).toarray()
            m = S.maximum(A, dims=dims).toarray()
        finally:
            S.set_option("reduce_tree", 0)
        assert np.array_equal(r1, r2)
        got[tree] = (r1, m)
    want = a.astype(np.float64).sum(axis=dims, keepdims=True)
    assert np.array_equal(got[1024][0].astype(np.float64), want) and np.array_equal(got[0][0].astype(np.float64), want)
    assert np.array_equal(got[1024][1], a.max(axis=dims, keepdims=True)) and np.array_equal(got[0][1], a.max(axis=dims, keepdims=True))


def test_two_level_fold_is_reproducible_and_accurate_on_real_data():
    rng = np.random.default_rng(5)
    a = rng.standard_normal((100, 50400)).astype(np.float32)
    A = dview(a)
    S.set_option("reduce_tree", 1024)   # off by default: measured slower than the second launch (profiles/r04_reduce_tree_ab.txt)
    try:
        runs = [S.sum(A, dims=(1,)).toarray() for _ in range(5)]
        for r in runs[1:]:
            assert np.array_equal(r, runs[0])                    # fold order is fixed by the lane layout, not by arrival order
        want = a.astype(np.float64).sum(axis=1, keepdims=True)
        assert np.allclose(runs[0], want, rtol=0, atol=2e-3 * np.sqrt(50400))
        # accumulate INTO a destination with an initop (the epilogue of the last shard applies it once)
        out = dview(np.full((100, 1), 2.0, dtype=np.float32))
        S.mapreducedim_(lambda x: x, "+", out, A)
        sync()
        assert np.allclose(out.toarray(), want + 2.0, rtol=0, atol=2e-3 * np.sqrt(50400))
    finally:
        S.set_option("reduce_tree", 0)


@pytest.mark.parametrize("shape", [(100, 90, 80), (257, 129, 65), (17, 33, 65, 31), (200, 300, 70), (999, 1001), (130, 70, 50, 9), (1400, 1500)])
@pytest.mark.parametrize("dt", [np.float64, np.float32, np.complex128])
def test_ragged_transposes_with_long_unit_dims(shape, dt):
    """Two-sided FLAT form with evenly cut leads (round 4): every permutation that moves dim 0, bit-exact vs NumPy; the same with
    the form switched off (TILED)."""
    import itertools
    import torch
    rng = np.random.default_rng(sum(shape))
    a = rng.standard_normal(shape).astype(dt)
    A = dview(a)
    n = len(shape)
    perms = [q for q in itertools.permutations(range(n)) if q[0] != 0]
    if n == 4:
        perms = [(3, 2, 1, 0), (3, 2, 0, 1), (1, 0, 2, 3), (2, 3, 0, 1), (1, 3, 0, 2)]
    hit = 0
    for q in perms:
        out = dview(np.zeros(tuple(shape[i] for i in q), dtype=dt))
        for v in (100, 0):   # 100: wherever the form applies (every array here has fewer than 8 MiB or well-filled tiles otherwise)
            S.set_option("flat2_long", v)
            try:
                plan = S.make_plan(lambda x: x, None, None, out.size, (out, A.permutedims(q)))
                hit += int("two-sided" in plan.describe() and v == 100)
                out.parent.zero_()
                plan.execute()
                torch.cuda.synchronize()
            finally:
                S.set_option("flat2_long", 80)
            assert np.array_equal(out.toarray(), np.transpose(a, q)), (shape, q, v, plan.describe())
    if shape == (257, 129, 65) and dt == np.float64:
        assert hit >= 2, hit


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_nary_maps_with_long_odd_unit_dims(dt):
    """`C .= C ./ 2 .+ 2 .* permutedims(A, p)` (TensorOperations' tensoradd!) on shapes whose unit-stride dims are long and odd: the
    two-sided FLAT form with cut leads reads the same-layout inputs at the destination's offsets."""
    import torch
    rng = np.random.default_rng(31)
    for shape, q in (((257, 129, 65), (1, 0, 2)), ((301, 303, 35), (2, 0, 1)), ((2049, 2051), (1, 0))):
        a = rng.integers(-50, 50, size=shape).astype(dt)
        c0 = rng.integers(-50, 50, size=tuple(shape[i] for i in q)).astype(dt)
        d0 = rng.integers(-50, 50, size=tuple(shape[i] for i in q)).astype(dt)
        A, C, D = dview(a), dview(c0), dview(d0)
        # (round 6: with element-aligned vectors TILED takes the odd extents below 32 MiB; this test is about the FLAT form)
        S.set_option("tiled_uavec", 0)
        try:
            plan = S.make_plan(lambda c, x, d: c / 2 + 2 * x - d, None, None, C.size, (C, C, A.permutedims(q), D))
        finally:
            S.set_option("tiled_uavec", 1)
        assert "two-sided" in plan.describe(), plan.describe()
        plan.execute()
        torch.cuda.synchronize()
        assert np.array_equal(C.toarray(), c0 / dt(2) + dt(2) * np.transpose(a, q) - d0), (shape, q)
        # ... and the default plan (TILED with element-aligned 16-byte accesses where it applies) computes the same
        C2 = dview(c0)
        plan2 = S.make_plan(lambda c, x, d: c / 2 + 2 * x - d, None, None, C2.size, (C2, C2, A.permutedims(q), D))
        plan2.execute()
        torch.cuda.synchronize()
        assert np.array_equal(C2.toarray(), c0 / dt(2) + dt(2) * np.transpose(a, q) - d0), (shape, q, plan2.describe())


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_two_sided_flat_pairs(dt):
    """The pair form of the two-sided FLAT kernel (forced with flat2_pair = 2): rows whose first element sits at an odd element address
    start with a single element, the others with a pair; plain and n-ary maps, group runs and cut leads."""
    import torch
    rng = np.random.default_rng(41)
    S.set_option("flat2_pair", 2)
    try:
        for shape, q in (((5, 300, 30, 7), (3, 2, 1, 0)), ((17, 33, 65, 31), (3, 2, 1, 0)), ((257, 129, 65), (2, 1, 0)), ((9, 70, 3, 90, 11), (4, 3, 2, 1, 0)),
                         ((6, 64, 64, 16, 5), (4, 3, 2, 1, 0))):
            a = rng.integers(-99, 99, size=shape).astype(dt)
            c0 = rng.integers(-99, 99, size=tuple(shape[i] for i in q)).astype(dt)
            A, C = dview(a), dview(c0)
            # an odd element offset on both sides: a sub-view that drops the first index of the LAST dim keeps the layout, shifts the parity
            plan = S.make_plan(lambda c, x: c - 2 * x, None, None, C.size, (C, C, A.permutedims(q)))
            plan.execute()
            torch.cuda.synchronize()
            assert np.array_equal(C.toarray(), c0 - dt(2) * np.transpose(a, q)), (shape, q, plan.describe())
    finally:
        S.set_option("flat2_pair", 1)
