"""GPU: round-6 kernel changes, each against NumPy (and bit-exact where the operation is a copy).

  * TILED: ragged extents run the lean kernel plus bounds checks (variant 1) instead of the everything-variant;
  * TILED: 16-byte accesses at element alignment -- odd extents, odd row strides, views that begin inside a vector -- with the partial
    vector at the end of a row moved element by element (options tiled_uavec);
  * REDUCE_PART, COL form: lane maps sized to the row (25 lanes x 10 rows for 100 Float32), any row count in the LDS fold;
  * the second pass of a split partial reduction loads its partials in batches of eight.
Reference semantics: src/mapreduce.jl:38-53 (map!), :55-96 (mapreducedim!), test sizes as in test/othertests.jl:68-107.
"""
import itertools

import numpy as np
import pytest

import strided_jl_amd as S
from strided_jl_amd import _lib as L

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


def host(view):
    return view.parent.cpu().numpy().reshape(view.size, order="F")


def cur():
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def sync():
    import torch
    torch.cuda.synchronize()


@pytest.fixture
def option():
    lib = L.load()
    saved = {}

    def setopt(name, value):
        if name not in saved:
            saved[name] = lib.smr_get_option(name.encode())
        L.check(lib.smr_set_option(name.encode(), value))

    yield setopt
    for k, v in saved.items():
        L.check(lib.smr_set_option(k.encode(), v))


RAGGED = [((999, 1001), (1, 0)), ((1001, 999), (1, 0)), ((100, 90, 80), (1, 0, 2)), ((100, 90, 80), (2, 1, 0)), ((90, 101, 33), (1, 2, 0)), ((7200, 100), (1, 0)),
          ((257, 129, 65), (2, 1, 0)), ((63, 65, 67), (2, 0, 1)), ((35, 1000), (1, 0)), ((1000, 35), (1, 0))]


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.int64])
@pytest.mark.parametrize("dims,perm", RAGGED)
def test_ragged_permutedims_bit_exact(dims, perm, dt, option):
    """Copies through every tiled variant: the destination must equal NumPy's transpose bit for bit, with and without the
    element-aligned vector path, and the plan says which family ran."""
    rng = np.random.default_rng(sum(dims) * 7 + sum(i * p for i, p in enumerate(perm)))
    a = (rng.standard_normal(dims) * 100).astype(dt) if dt != np.complex64 else (rng.standard_normal(dims) + 1j * rng.standard_normal(dims)).astype(dt)
    want = np.transpose(a, perm)
    for flat, ua in ((0, 1), (0, 0), (1, 1)):
        option("flat", flat)
        option("tiled_uavec", ua)
        A = dview(a)
        B = dview(np.zeros(want.shape, dtype=dt))
        plan = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(perm)))
        plan.execute(cur())
        sync()
        assert np.array_equal(host(B), want), f"{dims} {perm} {dt.__name__} flat={flat} tiled_uavec={ua}: {plan.describe()}"
        if not flat:
            assert "family=tiled" in plan.describe() or "family=stream" in plan.describe() or "family=generic" in plan.describe()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_unaligned_views_transposing_add(dt, option):
    """Views that begin inside a vector and have odd row strides: B[1:, :] .= A[2:, 1:]' .+ C[:-1, :-2] on (n, m) parents."""
    rng = np.random.default_rng(5)
    for (n, m) in ((1001, 777), (514, 515), (300, 301)):
        a = rng.standard_normal((m + 1, n + 2)).astype(dt)
        c = rng.standard_normal((n + 1, m + 2)).astype(dt)
        b = rng.standard_normal((n + 1, m)).astype(dt)
        for ua in (1, 0):
            option("tiled_uavec", ua)
            A, Cc, B = dview(a), dview(c), dview(b)
            dst = B.sview(slice(1, None), slice(None))
            x = A.sview(slice(1, None), slice(2, None)).permutedims((1, 0))
            y = Cc.sview(slice(0, n), slice(0, m))
            plan = S.make_plan(lambda u, v: u + v, None, None, dst.size, (dst, x, y))
            plan.execute(cur())
            sync()
            want = b.copy()
            want[1:, :] = a[1:, 2:].T + c[:n, :m]
            assert np.array_equal(host(B), want), f"({n},{m}) {dt.__name__} tiled_uavec={ua}: {plan.describe()}"


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.int64])
@pytest.mark.parametrize("dims", [(100, 90, 80, 7), (50, 33, 40), (36, 35, 34, 3), (250, 17, 19)])
def test_partial_sums_every_dim_subset(dims, dt, option):
    """sum over every subset of dims (mapreducedim! with op = +, initop = zero): exact lane maps on and off; integers exactly,
    floats to sqrt(eps) relative to the sum of magnitudes."""
    rng = np.random.default_rng(11)
    a = rng.integers(-1000, 1000, size=dims).astype(dt) if dt == np.int64 else rng.standard_normal(dims).astype(dt)
    mag = np.abs(a.astype(np.float64))
    for exact in (1, 0):
        option("reduce_col_exact", exact)
        for k in range(1, len(dims)):
            for rd in itertools.combinations(range(len(dims)), k):
                A = dview(a)
                oshape = tuple(1 if d in rd else n for d, n in enumerate(dims))
                out = dview(np.zeros(oshape, dtype=dt))
                plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
                plan.execute(cur())
                sync()
                got = host(out)
                if dt == np.int64:
                    assert np.array_equal(got, a.sum(axis=rd, keepdims=True)), f"{dims} dims={rd} exact={exact}: {plan.describe()}"
                else:
                    want = a.astype(np.float64).sum(axis=rd, keepdims=True)
                    tol = np.sqrt(np.finfo(dt).eps) * mag.sum(axis=rd, keepdims=True)
                    assert np.all(np.abs(got.astype(np.float64) - want) <= tol), f"{dims} dims={rd} {dt.__name__} exact={exact}: {plan.describe()}"


def test_exact_lane_map_is_planned():
    """Rows of 100 Float32 / Float64 get 25 x 10 / 50 x 5 lanes; rows of 128 keep the power of two."""
    for dt, lanes in ((np.float32, "lanes=25x10"), (np.float64, "lanes=50x5")):
        dims = (100, 90, 80, 7)
        A = dview(np.zeros(dims, dtype=dt))
        out = dview(np.zeros((100, 1, 1, 1), dtype=dt))
        d = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A)).describe()
        assert "form=col" in d and lanes in d, d
    dims = (128, 90, 80)
    A = dview(np.zeros(dims, dtype=np.float32))
    out = dview(np.zeros((128, 1, 1), dtype=np.float32))
    d = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A)).describe()
    assert "form=col" in d and "lanes=" not in d, d


@pytest.mark.parametrize("op,npop", [("max", np.max), ("min", np.min), ("*", np.prod)])
def test_split_reductions_other_ops(op, npop, option):
    """The batched second pass with the other reduction operators (the unused slots of its last batch must not reach the result)."""
    rng = np.random.default_rng(3)
    dims = (100, 9, 4001)
    a = (1.0 + 1e-4 * rng.standard_normal(dims)).astype(np.float64)
    A = dview(a)
    for rd in ((1, 2), (0, 2), (2,)):
        res = {"max": S.maximum, "min": S.minimum, "*": S.prod}[op](A, dims=rd)
        sync()
        want = npop(a, axis=rd, keepdims=True)
        got = S.Array(res)
        assert np.allclose(got.reshape(want.shape), want, rtol=1e-10), f"{op} dims={rd}"


PERMS4 = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]


@pytest.mark.parametrize("dt", [np.float64, np.complex64])
@pytest.mark.parametrize("n", [32, 24, 40])
def test_orbit_pair_form_bit_exact(n, dt, option):
    """ORBIT, PAIR form (two slot sets per workgroup, unit-axis neighbours in the lane pairs; 4^4 cubes of 8-byte elements), forced
    for plain-store launches too (orbit_pair = 2): the 4-way permuted sum (README.md:104 of the reference), three and two views of
    the same rotation group, a conjugated view, and the in-place update -- each bit-identical to NumPy and to the one-orbit form."""
    rng = np.random.default_rng(n)
    a = rng.integers(-999, 999, size=(n,) * 4).astype(dt)
    if dt == np.complex64:
        a = (a + 1j * rng.integers(-999, 999, size=(n,) * 4)).astype(dt)
    views = lambda A, idx: tuple(A.permutedims(PERMS4[i]) for i in idx)
    cases = [((0, 1, 2, 3), lambda w, x, y, z: w + x + y + z, lambda v: ((v[0] + v[1]) + v[2]) + v[3]),
             ((0, 1, 3), lambda w, x, y: w + x + y, lambda v: (v[0] + v[1]) + v[2]),
             ((0, 2), lambda w, x: w + x, lambda v: v[0] + v[1])]
    for idx, f, ref in cases:
        want = ref([np.transpose(a, PERMS4[i]) for i in idx])
        got = {}
        for pair in (2, 0):
            option("orbit_pair", pair)
            A, Cc = dview(a), dview(np.zeros_like(a))
            plan = S.make_plan(f, None, None, A.size, (Cc,) + views(A, idx))
            d = plan.describe()
            plan.execute(cur())
            sync()
            got[pair] = host(Cc)
            assert np.array_equal(got[pair], want), f"n={n} {dt.__name__} views={idx} orbit_pair={pair}: {d}"
            if n % 8 == 0 and pair == 2 and "family=orbit" in d and "tile=d0:4" in d:
                assert "pair_grid=" in d, d
            # in place: the destination is the buffer
            A2 = dview(a)
            plan2 = S.make_plan(f, None, None, A2.size, (A2,) + views(A2, idx))
            if "family=orbit" in plan2.describe():  # (only the orbit family reads a whole orbit before it writes it: aliasing is safe there)
                plan2.execute(cur())
                sync()
                assert np.array_equal(host(A2), want), f"in place, n={n} {dt.__name__} views={idx} orbit_pair={pair}: {plan2.describe()}"
    if dt == np.complex64:  # a conjugated view
        option("orbit_pair", 2)
        A, Cc = dview(a), dview(np.zeros_like(a))
        plan = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (Cc, A, A.permutedims(PERMS4[1]).conj(), A.permutedims(PERMS4[2]), A.permutedims(PERMS4[3]).conj()))
        plan.execute(cur())
        sync()
        want = ((a + np.conj(np.transpose(a, PERMS4[1]))) + np.transpose(a, PERMS4[2])) + np.conj(np.transpose(a, PERMS4[3]))
        assert np.array_equal(host(Cc), want), plan.describe()


def test_orbit_pair_form_runs_in_recorded_sequences():
    """The replayed bench step (smr_seq: write-through launches) takes the PAIR form by itself and stays exact."""
    n = 32
    rng = np.random.default_rng(9)
    a = rng.standard_normal((n,) * 4)
    A, B, Cc = dview(a), dview(np.zeros_like(a)), dview(np.zeros_like(a))
    p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    p3 = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in PERMS4))
    seq = S.Sequence().add(p2).add(p3)
    seq.run(5, cur())
    seq.wait()
    sync()
    assert np.array_equal(host(B), np.transpose(a, (3, 2, 1, 0)))
    at = lambda p: np.transpose(a, p)
    assert np.array_equal(host(Cc), ((at(PERMS4[0]) + at(PERMS4[1])) + at(PERMS4[2])) + at(PERMS4[3]))


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
@pytest.mark.parametrize("dims,perm", [((7200, 100), (1, 0)), ((100, 7200), (1, 0)), ((104, 3000), (1, 0)), ((100, 90, 80), (1, 0, 2)), ((100, 90, 80), (2, 1, 0)),
                                       ((65, 129, 33), (2, 1, 0)), ((127, 1000), (1, 0)), ((1000, 127), (1, 0))])
def test_flat_wide_rows_bit_exact(dims, perm, dt, option):
    """One-sided FLAT form with whole rows of 65..128 elements (flat_wide = 2 forces it wherever it applies; the planner's own rule
    takes it for matrices of 32 MiB and more): copies and an n-ary map, bit-identical to NumPy."""
    rng = np.random.default_rng(sum(dims) + 3 * len(perm))
    a = rng.integers(-999, 999, size=dims).astype(dt)
    want = np.transpose(a, perm)
    hit = 0
    for fw in (2, 0):
        option("flat_wide", fw)
        A = dview(a)
        B = dview(np.zeros(want.shape, dtype=dt))
        plan = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(perm)))
        plan.execute(cur())
        sync()
        assert np.array_equal(host(B), want), f"{dims} {perm} {dt.__name__} flat_wide={fw}: {plan.describe()}"
        hit += int(fw == 2 and "family=flat" in plan.describe())
        c0 = rng.integers(-9, 9, size=want.shape).astype(dt)
        Cc = dview(c0)
        plan = S.make_plan(lambda c, x: c + 2 * x, None, None, Cc.size, (Cc, Cc, A.permutedims(perm)))
        plan.execute(cur())
        sync()
        assert np.array_equal(host(Cc), c0 + dt(2) * want), f"n-ary {dims} {perm} {dt.__name__} flat_wide={fw}: {plan.describe()}"
    if dt != np.complex64 and dims in ((7200, 100), (127, 1000), (1000, 127)):
        assert hit == 1, "flat_wide = 2 did not select the FLAT family"


def test_flat_wide_is_planned_for_big_matrices_only():
    def fam(dims, dt):
        A = dview(np.zeros(dims, dtype=dt))
        B = dview(np.zeros(dims[::-1], dtype=dt))
        return S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims((1, 0)))).describe()
    assert "family=flat" in fam((100, 200000), np.float32) and "run=100x1" in fam((100, 200000), np.float32)
    assert "family=tiled" in fam((7200, 100), np.float64)


@pytest.mark.parametrize("shape,perms", [
    ((16, 16, 5, 16, 16), [(0, 1, 2, 3, 4), (1, 3, 2, 4, 0), (3, 4, 2, 0, 1), (4, 0, 2, 1, 3)]),   # a batch dim between the tiled ones
    ((32, 32, 32, 32), [(0, 1, 2, 3), (3, 0, 1, 2), (2, 3, 0, 1), (1, 2, 3, 0)]),                   # the rotation the other way
    ((32, 32, 32, 32), [(0, 1, 2, 3), (1, 0, 3, 2), (2, 3, 0, 1), (3, 2, 1, 0)]),                   # Klein four-group, not cyclic
    ((20, 20, 20, 20), [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]),                   # 5 tiles per dim: an odd one out
    ((32, 32, 32, 32, 3), [(0, 1, 2, 3, 4), (1, 2, 3, 0, 4), (2, 3, 0, 1, 4), (3, 0, 1, 2, 4)])])  # trailing batch dim (8^4 cubes: no PAIR form)
def test_orbit_pair_form_other_groups_and_batch_dims(shape, perms, option):
    rng = np.random.default_rng(len(shape) + shape[0])
    for dt in (np.float64, np.complex64):
        a = rng.integers(-99, 99, size=shape).astype(dt)
        want = None
        for p in perms:
            t = np.transpose(a, p)
            want = t if want is None else want + t
        for pair in (1, 0):
            option("orbit_pair", pair)
            A, Cc = dview(a), dview(np.zeros_like(a))
            plan = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))
            plan.execute(cur())
            sync()
            assert np.array_equal(host(Cc), want), f"{shape} {dt.__name__} orbit_pair={pair}: {plan.describe()}"
