"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the NumPy restatement
of the reference's assertions, on the full case list of tests/cases.py.  Bit-exact where the
reference compares with `==` (copies, permutes, conj, integer-valued matmul, exact arithmetic
maps); `isapprox` with rtol = sqrt(eps) -- the reference's own tolerance -- elsewhere
(transcendentals differ between OCML and libm, reductions differ in summation order)."""
import sys

import numpy as np
import pytest

import cases
import oraclelib
import strided_jl_amd as S
from util import fview, rtol

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


def _isapprox(r, e, tol):
    r = np.asarray(r).astype(np.complex128).ravel()
    e = np.asarray(e).astype(np.complex128).ravel()
    return np.linalg.norm(r - e) <= tol * max(np.linalg.norm(r), np.linalg.norm(e))


def _oracle(name, monkeypatch):
    def funnel(f, op, initop, dims, arrays):
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, 1)
        return arrays[0]

    with monkeypatch.context() as m:
        m.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
        return cases.run_case(name, fview)


@pytest.mark.parametrize("name", cases.NAMES)
def test_hip_matches_oracle_and_numpy(name, monkeypatch):
    import torch
    res_o, exp, exact = _oracle(name, monkeypatch)
    res_d, exp_d, _ = cases.run_case(name, dview)
    torch.cuda.synchronize()
    assert len(res_d) == len(res_o) == len(exp)
    for i, (d, o, e) in enumerate(zip(res_d, res_o, exp)):
        d, o, e = np.asarray(d), np.asarray(o), np.asarray(e)
        assert d.shape == o.shape == e.shape, f"{name}[{i}] shape"
        assert d.dtype == o.dtype, f"{name}[{i}] dtype {d.dtype} vs oracle {o.dtype}"
        if exact:
            assert np.array_equal(d, o), f"{name}[{i}]: HIP result is not bit-identical to the oracle"
            assert np.array_equal(d, e), f"{name}[{i}]: HIP result is not bit-identical to NumPy"
        else:
            tol = max(rtol(x.dtype if np.issubdtype(x.dtype, np.inexact) else np.float64) for x in (d, e))
            assert _isapprox(d, o, tol), f"{name}[{i}]: HIP vs oracle"
            assert _isapprox(d, e, tol), f"{name}[{i}]: HIP vs NumPy"


def test_arithmetic_maps_are_bit_identical_to_the_oracle(monkeypatch):
    """+ - * / maps carry no libm: with -ffp-contract=off on both sides the HIP results must
    equal the oracle's bit for bit (scale, axpy, axpby, symmetrise, 4-way sum)."""
    import torch
    rng = np.random.default_rng(7)
    for T in cases.FLOATS:
        for N in (2, 3, 4):
            dims = (24 // N * 2,) * N
            R = [cases._rand(rng, dims, T) for _ in range(3)]
            P = [cases._randperm(rng, N) for _ in range(3)]

            def run(mk):
                B = [mk(r).permutedims(p) for r, p in zip(R, P)]
                out = []
                S.rmul_(B[0], 0.75); out.append(B[0].toarray())
                S.axpy_(1.25, B[0], B[1]); out.append(B[1].toarray())
                S.axpby_(0.3, B[1], -1.7, B[2]); out.append(B[2].toarray())
                D = mk(np.zeros(B[0].size, dtype=T))
                D.assign((B[0] + B[1]) / 2); out.append(D.toarray())
                D.assign(B[0] + B[1] + B[2] + B[0]); out.append(D.toarray())
                D.assign(B[0] * B[1] - B[2] / 3); out.append(D.toarray())
                return out

            def funnel(f, op, initop, dims_, arrays):
                p, keep = S.build_problem(f, op, initop, dims_, arrays, stream=0)
                oraclelib.mapreduce(p, 1)
                return arrays[0]

            with monkeypatch.context() as m:
                m.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
                want = run(fview)
            got = run(dview)
            torch.cuda.synchronize()
            for i, (g, w) in enumerate(zip(got, want)):
                if np.issubdtype(np.dtype(T), np.complexfloating) and i == 5:
                    # complex division: Smith's algorithm on the device vs libgcc's on the host
                    assert _isapprox(g, w, rtol(T))
                else:
                    assert np.array_equal(g, w), f"{np.dtype(T).name} N={N} step {i}"


@pytest.mark.parametrize("T", [np.float32, np.float64, np.complex128])
@pytest.mark.parametrize("shape", [(200, 200), (1000, 1000), (20, 20, 20, 20), (48, 16, 48, 16), (33, 65, 33)])
def test_aliased_permuted_inputs_with_orbit_tile_order(shape, T):
    """Inputs that are permuted views of ONE buffer take the orbit-major tile order (in-kernarg
    table for small grids, device table for big ones, ragged edge tiles, padding workgroups):
    results must equal NumPy bit for bit, with the option on and off."""
    import torch
    rng = np.random.default_rng(11)
    a = cases._rand(rng, shape, T)
    N = len(shape)
    if N == 2:
        perms = [(0, 1), (1, 0)]
    elif N == 3:
        perms = [(0, 1, 2), (2, 1, 0)]
    else:
        perms = [(0, 1, 2, 3), (2, 1, 0, 3), (0, 3, 2, 1), (2, 3, 0, 1)]
    want = a.transpose(perms[0]).copy()
    for q in perms[1:]:
        want = want + a.transpose(q)
    for orbit, order in ((1, 1), (0, 1), (0, 0)):
        S.set_option("orbit", orbit)
        S.set_option("tile_order", order)
        try:
            A = dview(a)
            B = A.similar()
            views = [A.permutedims(q) for q in perms]
            e = views[0]
            for v in views[1:]:
                e = e + v
            B.assign(e)
            plan = S.make_plan((lambda *xs: sum(xs[1:], xs[0])), None, None, B.size, (B, *views))
            d = plan.describe()
            if orbit == 0 and shape == (33, 65, 33) and np.dtype(T).itemsize < 16:
                # round 3: one input laid out like the destination + one transposed, unit-stride dims of 33 elements: n-ary two-sided FLAT
                assert "family=flat" in d and "two-sided" in d, d
            elif orbit == 0:
                assert "family=tiled" in d
                assert ("order=orbits" in d) == (order == 1 and bool(plan.tile_order()))
            else:
                # FAM_ORBIT where the sizes allow it (power-of-two divisible unit dims, enough orbits)
                # (200 = 8 * 25, 1000 = 8 * 125, 20, 33, 65: no tile of >= 256 elements divides them -> classic kernel, ragged tiles)
                assert ("family=orbit" in d) == (shape == (48, 16, 48, 16)), d
            torch.cuda.synchronize()
            assert np.array_equal(B.toarray(), want), f"{shape} {np.dtype(T).name} orbit={orbit} tile_order={order}: {d}"
        finally:
            S.set_option("tile_order", 1)
            S.set_option("orbit", 1)


def test_mapreduce_scalar_returns_the_complete_reduction_to_the_host():
    """smr_mapreduce_scalar = the synchronous tail of `_mapreduce` (src/mapreduce.jl:70-71)."""
    import ctypes as C
    from strided_jl_amd import _lib as L
    rng = np.random.default_rng(3)
    a = cases._rand(rng, (37, 41, 13), np.float64)
    A = dview(a).permutedims((2, 0, 1))
    out = A.similar(np.float64, (1,))
    S.copyto_(out, 0.0)
    p, keep = S.build_problem(S.fn.abs2, "+", None, A.size, S.promoteshape(A.size, out.sreshape((1, 1, 1)), A),
                              stream=0)
    host = C.c_double(-1.0)
    L.check(L.load().smr_mapreduce_scalar(C.byref(p), C.byref(host)))
    assert abs(host.value - float((a * a).sum())) <= 1e-12 * float((a * a).sum())
    assert out.item() == host.value
    # a destination with more than one element is refused
    big = A.similar()
    p, keep = S.build_problem(S.fn.abs2, "+", None, A.size, (big, A), stream=0)
    assert L.load().smr_mapreduce_scalar(C.byref(p), C.byref(host)) == L.SMR_EINVAL


@pytest.mark.parametrize("T", cases.FLOATS)
def test_runtime_compiled_functors_equal_the_interpreter_bit_for_bit(T):
    """f-programs without a native functor are compiled by hiprtc into straight-line code over the
    same primitives the bytecode interpreter calls: both paths must agree exactly, in every kernel
    family (stream, tiled, tiled+ragged, generic, complete and partial reductions, mixed types)."""
    import torch
    fn = S.fn
    rng = np.random.default_rng(5)
    a, b, c = (cases._rand(rng, (48, 40, 12), T) for _ in range(3))
    cplx = np.issubdtype(np.dtype(T), np.complexfloating)

    def run():
        A, B, C = dview(a), dview(b), dview(c)
        out = []
        D = A.similar()
        D.assign(A * 2 + B / 3 - 1); out.append(D.toarray())                                  # stream
        D.assign(fn.sqrt(fn.abs(A)) * B - fn.exp(C * 0.25)); out.append(D.toarray())
        E = dview(np.zeros((40, 48, 12), dtype=T))
        E.assign(A.permutedims((1, 0, 2)) * B.permutedims((1, 0, 2)) - C.permutedims((1, 0, 2)) / 7); out.append(E.toarray())  # tiled
        F = dview(np.zeros((12, 40, 48), dtype=T))
        F.assign(fn.abs2(A.permutedims((2, 1, 0))) - B.permutedims((2, 1, 0)) * 3); out.append(F.toarray())  # tiled, 3 axes, ragged
        if not cplx:
            D.assign(fn.select(A < B, A - C, B * C)); out.append(D.toarray())
        G = A.sview(slice(0, 48, 2), slice(None), slice(None))                                 # generic (stride 2)
        H = dview(np.zeros(G.size, dtype=T))
        S.map_(lambda x, y: x * y - x, H, G, B.sview(slice(0, 48, 2), slice(None), slice(None))); out.append(H.toarray())
        out.append(np.asarray(S.mapreduce(lambda x: fn.sin(x) * x, "+", A)))                  # reduce_all
        out.append(S.mapreduce(lambda x: x * x - 1, "+", A, dims=(1,)).toarray())              # reduce_part
        Dd = dview(np.zeros((48, 40, 12), dtype=np.complex128 if cplx else np.float64))
        Dd.assign(A * B - 0.5); out.append(Dd.toarray())                                       # mixed
        torch.cuda.synchronize()
        return out

    c0, f0 = S.get_option("jit_compiles"), S.get_option("jit_failures")
    S.set_option("jit", 1)
    jit = run()
    assert S.get_option("jit_failures") == f0, "hiprtc failed on the GPU box"
    assert S.get_option("jit_compiles") + S.get_option("jit_hits") > c0
    S.set_option("jit", 0)
    try:
        interp = run()
    finally:
        S.set_option("jit", 1)
    for i, (x, y) in enumerate(zip(jit, interp)):
        assert np.array_equal(x, y, equal_nan=True), f"{np.dtype(T).name} expression {i}: JIT differs from the interpreter"
    # and both agree with NumPy on the pure-arithmetic ones
    assert np.array_equal(jit[0], a * 2 + b / 3 - 1) or cplx


@pytest.mark.parametrize("T", [np.float32, np.float64, np.complex64])
@pytest.mark.parametrize("shape,dims", [((300, 200), (0,)), ((300, 200), (1,)), ((64, 33, 20, 9), (0, 2)), ((64, 33, 20, 9), (1, 3)),
                                        ((12, 70000), (1,)), ((70000, 12), (0,)), ((40, 40, 40), (0, 1)), ((37, 5, 41), (1,)),
                                        ((16, 8, 6, 4, 10), (1, 2, 4))])
def test_partial_reductions_vectorised_forms_match_the_general_kernel_and_numpy(shape, dims, T):
    """sum/mapreduce over some dims: the ROW form (inputs contiguous along a reduced dim), the COL
    form (contiguous along a kept dim), their split two-pass variants (few outputs, long
    reductions) and ragged extents, against the general kernel and NumPy in float64."""
    import torch
    fn = S.fn
    rng = np.random.default_rng(17)
    a = cases._rand(rng, shape, T)
    b = cases._rand(rng, shape, T)
    truth = (np.sin(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)) * b).sum(axis=dims, keepdims=True)
    tol = rtol(T)
    res = {}
    for kind in (-1, 0):
        S.set_option("reduce_part_kind", kind)
        try:
            A, B = dview(a), dview(b)
            out = A.similar(size=tuple(1 if d in dims else n for d, n in enumerate(shape)))
            S.copyto_(out, 1.5)
            # out = out*2 + sum(sin(a)*b): exercises initop-once on top of the reduction
            S._mapreducedim_(lambda x, y: fn.sin(x) * y, "+", ("scale", 2.0), shape, (out, A, B))
            plan = S.make_plan(lambda x, y: fn.sin(x) * y, "+", ("scale", 2.0), shape, S.promoteshape(shape, out, A, B))
            torch.cuda.synchronize()
            res[kind] = (out.toarray(), plan.describe())
        finally:
            S.set_option("reduce_part_kind", -1)
    assert "form=general" in res[0][1]
    assert "form=row" in res[-1][1] or "form=col" in res[-1][1], res[-1][1]
    for kind, (got, d) in res.items():
        assert _isapprox(got, truth + 3.0, tol), f"{shape} dims={dims} {np.dtype(T).name}: {d}"


def test_c_level_comm_single_rank_roundtrip():
    """smr_comm_*: a real one-rank RCCL communicator on the GPU box; smr_mapreduce_sharded then
    equals smr_mapreduce (the multi-rank decomposition itself is covered by the 2-rank gloo test
    of the Python twin, tests/test_distributed_cpu.py)."""
    import ctypes as C
    from strided_jl_amd import _lib as L
    lib = L.load()
    uid = C.create_string_buffer(128)
    L.check(lib.smr_comm_unique_id(uid, 128))
    assert any(uid.raw), "ncclGetUniqueId returned zeros"
    L.check(lib.smr_comm_init(1, 0, uid, 128))
    try:
        rank, n = C.c_int(-1), C.c_int(-1)
        lib.smr_comm_rank(C.byref(rank), C.byref(n))
        assert (rank.value, n.value) == (0, 1)
        rng = np.random.default_rng(9)
        a = cases._rand(rng, (33, 20, 17), np.float64)
        A = dview(a)
        out = A.similar(size=(33, 1, 1))
        S.copyto_(out, 2.0)
        p, keep = S.build_problem(S.fn.abs2, "+", "identity", A.size, S.promoteshape(A.size, out, A), stream=0)
        L.check(lib.smr_mapreduce_sharded(C.byref(p)))
        assert _isapprox(out.toarray(), (a * a).sum(axis=(1, 2), keepdims=True) + 2.0, 1e-12)
    finally:
        L.check(lib.smr_comm_destroy())


@pytest.mark.parametrize("T", [np.float32, np.complex128, np.int16])
def test_strided_rows_take_the_stream_family(T):
    """Views with a step along the fastest dim (A[1:2:end, :], reversed ranges, a destination that
    is itself a strided view) run in the STREAM family's element-wise form, not in the
    one-thread-per-element fallback; results equal NumPy exactly."""
    import torch
    rng = np.random.default_rng(23)
    a = cases._rand(rng, (400, 37, 5), T) if T != np.int16 else rng.integers(-1000, 1000, (400, 37, 5)).astype(T)
    b = cases._rand(rng, (200, 37, 5), T) if T != np.int16 else rng.integers(-1000, 1000, (200, 37, 5)).astype(T)
    A, B = dview(a), dview(b)
    D = dview(np.zeros((200, 37, 5), dtype=T))
    src = A.sview(slice(1, 400, 2), slice(None), slice(None))
    rev = B.sview(slice(199, None, -1), slice(None), slice(None))
    if T == np.int16:
        S.copy_(D, src)
        d = S.make_plan(lambda x: x, None, None, D.size, (D, src)).describe()
        assert "family=stream" in d and "vec=1" in d
        torch.cuda.synchronize()
        assert np.array_equal(D.toarray(), a[1::2])
        S.copy_(D, rev)
        torch.cuda.synchronize()
        assert np.array_equal(D.toarray(), b[::-1])
        return
    D.assign(src * 2 + rev)
    d = S.make_plan(lambda x, y: x * 2 + y, None, None, D.size, (D, src, rev)).describe()
    assert "family=stream" in d and "vec=1" in d, d
    torch.cuda.synchronize()
    assert np.array_equal(D.toarray(), a[1::2] * 2 + b[::-1])
    # strided destination: every third row of a bigger array
    big = dview(np.zeros((600, 37, 5), dtype=T))
    dst = big.sview(slice(0, 600, 3), slice(None), slice(None))
    dst.assign(B - src)
    torch.cuda.synchronize()
    want = np.zeros((600, 37, 5), dtype=T)
    want[0::3] = b - a[1::2]
    assert np.array_equal(big.toarray(), want)


def test_prepare_then_capture_into_a_hip_graph_without_a_warm_up_execution():
    """smr_plan_prepare builds the tables / compiles the kernel / allocates scratch; a plan prepared that
    way can be captured into a hipGraph straight away (no synchronous upload inside the capture)."""
    import torch
    rng = np.random.default_rng(41)
    a = cases._rand(rng, (96, 80), np.float64)
    A = dview(a)
    B = dview(np.zeros((80, 96)))
    out = dview(np.zeros((96, 1)))
    plans = [S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims((1, 0)))),                 # tiled, native functor
             S.make_plan(lambda x: x * x - x / 3, None, None, B.size, (B, A.permutedims((1, 0)))),     # tiled, runtime-compiled
             S.make_plan(lambda x: x, "+", "zero", A.size, S.promoteshape(A.size, out, A))]            # partial reduction
    wants = [a.T, a.T * a.T - a.T / 3, a.sum(axis=1, keepdims=True)]
    for plan, want, dst in zip(plans, wants, (B, B, out)):
        plan.prepare()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                plan.execute(int(torch.cuda.current_stream().cuda_stream))
        g.replay()
        torch.cuda.synchronize()
        assert _isapprox(dst.toarray(), want, 1e-13), plan.describe()


def test_every_kernel_family_is_exercised():
    """Plans for representative problems pick the intended family (guards against a silent
    fallback to the generic kernel)."""
    x = dview(np.zeros((32, 32, 32, 32)))
    y = x.similar()
    d = S.make_plan(lambda v: v, None, None, x.size, (y, x.permutedims((3, 2, 1, 0)))).describe()
    assert "family=tiled" in d and "f=ident" in d
    d = S.make_plan(lambda a, b: (a + b) / 2, None, None, (32, 32 ** 3), (y.sreshape((32, 32 ** 3)), x.sreshape((32, 32 ** 3)), x.sreshape((32, 32 ** 3)))).describe()
    assert "family=stream" in d
    o = x.similar(size=(1,))
    d = S.make_plan(S.fn.abs2, "+", None, x.size, S.promoteshape(x.size, o.sreshape((1, 1, 1, 1)), x)).describe()
    assert "family=reduce_all" in d and "f=abs2" in d
    o = x.similar(size=(32, 1, 32, 1))
    d = S.make_plan(S.fn.sin, "+", None, x.size, S.promoteshape(x.size, o, x)).describe()
    assert "family=reduce_part" in d
    # round 3: ORBIT (permuted views of one buffer) and FLAT (short leading dims that are not powers of two)
    ps = [x.permutedims(q) for q in [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]]
    assert "family=orbit" in S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, x.size, (y, *ps)).describe()
    img = dview(np.zeros((640, 480, 3)))
    out = dview(np.zeros((3, 480, 640)))
    assert "family=flat" in S.make_plan(lambda v: v, None, None, out.size, (out, img.permutedims((2, 1, 0)))).describe()


@pytest.mark.parametrize("T", [np.float32, np.float64, np.complex64, np.int32, np.int64])
def test_flat_family_image_layouts_are_bit_exact(T):
    """Round 3: planar <-> interleaved channels and the transposition of 3-element groups (FLAT family, all three forms),
    scaled copies included; integer types move as bit copies / in the integer class."""
    import torch
    rng = np.random.default_rng(11)
    H, W, Cn = 160, 202, 3
    if np.issubdtype(np.dtype(T), np.integer):
        a = rng.integers(-1000, 1000, size=(W, H, Cn)).astype(T)
    elif np.issubdtype(np.dtype(T), np.complexfloating):
        a = (rng.standard_normal((W, H, Cn)) + 1j * rng.standard_normal((W, H, Cn))).astype(T)
    else:
        a = rng.standard_normal((W, H, Cn)).astype(T)
    A = dview(a)
    seen = set()
    for q in ((2, 1, 0), (2, 0, 1), (1, 0, 2)):          # (3,H,W), (3,W,H), (H,W,3)
        B = dview(np.zeros(tuple(a.shape[i] for i in q), dtype=T))
        plan = S.make_plan(lambda v: v, None, None, B.size, (B, A.permutedims(q)))
        seen.add(plan.describe().split()[0])
        S.permutedims_(B, A, q)
        torch.cuda.synchronize()
        assert np.array_equal(B.toarray(), a.transpose(q)), (q, plan.describe())
        # and back, with a scale
        C2 = dview(np.zeros(a.shape, dtype=T))
        inv = tuple(int(i) for i in np.argsort(q))
        S.map_(lambda v: v * 3, C2, B.permutedims(inv))
        torch.cuda.synchronize()
        assert np.array_equal(C2.toarray(), a * T(3)), (q, "back")
    b = np.asfortranarray(a.transpose(2, 0, 1))          # (3, W, H): transposition of 3-element groups
    Bv = dview(b)
    D = dview(np.zeros((Cn, H, W), dtype=T))
    d = S.make_plan(lambda v: v, None, None, D.size, (D, Bv.permutedims((0, 2, 1)))).describe()
    S.permutedims_(D, Bv, (0, 2, 1))
    torch.cuda.synchronize()
    assert np.array_equal(D.toarray(), b.transpose(0, 2, 1)), d
    assert "family=flat" in seen and "shared-lead" in d


@pytest.mark.parametrize("T", [np.float64, np.float32, np.complex128, np.int64])
def test_one_launch_split_reductions_never_fold_stale_partials(T):
    """Round 3: a split reduction folds its partials inside the launch (the workgroup that arrives last at the group's
    counter reads what the others published write-through).  The per-XCD L2s are not coherent and a CU's L1 is never
    refreshed by other CUs' stores, so a broken hand-off shows as partials of the PREVIOUS launch: the input changes
    every launch here, the expected sums are exact, other kernels run on a second stream to make the load uneven, and
    the counters must be back at zero for the next launch (several hundred back-to-back launches of ONE plan)."""
    import torch
    lib = S._lib.load()
    keep = lib.smr_get_option(b"reduce_single")
    S._lib.check(lib.smr_set_option(b"reduce_single", 1 << 20))   # the one-launch form whatever the number of chunks
    try:
        _stale_partials_body(T, torch)
    finally:
        S._lib.check(lib.smr_set_option(b"reduce_single", keep))


def _stale_partials_body(T, torch):
    dims = (96, 50, 40, 7)
    a = np.ones(dims, dtype=T)
    A = dview(a)
    outs = []
    for rd in ((1, 2, 3), (0, 1, 2), (0, 1, 2, 3), (2, 3)):
        odims = tuple(1 if d in rd else n for d, n in enumerate(dims))
        out = A.similar(size=odims)
        plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
        d = plan.describe()
        assert "reduce" in d
        outs.append((rd, odims, out, plan, d))
    assert any("split=" in d and "split=1 " not in d for (_, _, _, _, d) in outs if "reduce_part" in d)
    big = torch.randn(1 << 24, device="cuda")
    side = torch.cuda.Stream()
    cur = lambda: int(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    hist = []
    for it in range(1, 161):
        A.parent.fill_(it)
        if it % 3 == 0:
            with torch.cuda.stream(side):
                big.mul_(1.0001)
        for (rd, odims, out, plan, d) in outs:
            plan.execute(cur())
        hist.append([out.parent.clone() for (_, _, out, _, _) in outs])   # queued behind the reductions: launches stay back to back
        if it % 40 and it > 8:
            continue              # the first launches are checked one by one, then 40 at a time without a host sync in between
        torch.cuda.synchronize()
        for j, snap in enumerate(hist):
            i0 = it - len(hist) + 1 + j
            for (rd, odims, out, plan, d), got in zip(outs, snap):
                cnt = int(np.prod([dims[i] for i in rd]))
                got = got.cpu().numpy()
                assert np.all(got == T(i0 * cnt)), (i0, rd, d, got.ravel()[:4], i0 * cnt)
        hist = []


def test_plain_array_rule_upload_on_the_device():
    """a5 (default rule 'upload'): a broadcast that mixes device views with a plain host array uploads the array, runs on the GPU and
    hands the out-of-place result back as a plain array (src/broadcast.jl:11-18, test/othertests.jl:64)."""
    import torch
    from strided_jl_amd.broadcast import set_plain_array_rule
    fn = S.fn
    rng = np.random.default_rng(5)
    for T in (np.float32, np.float64, np.complex64, np.complex128):
        R1, R2, R3 = (rng.random(sh).astype(T) for sh in ((10,), (10, 10), (10, 10, 10)))
        B1 = S.StridedView(torch.from_numpy(R1).cuda())
        B2 = S.StridedView(torch.from_numpy(R2).cuda()).permutedims((1, 0))
        B3 = S.StridedView(torch.from_numpy(R3).cuda()).permutedims((2, 0, 1))
        A3 = B3.toarray()
        old = set_plain_array_rule("error")
        try:
            with pytest.raises(TypeError):
                B2.adjoint() * A3
        finally:
            set_plain_array_rule(old)
        old = set_plain_array_rule("upload")
        try:
            got = (B2.adjoint() * A3 - fn.max(fn.abs(B1), fn.real(B3))).materialize()
            assert isinstance(got, np.ndarray)
            a1, a2 = B1.toarray(), B2.toarray()
            want = a2.conj().T[:, :, None] * A3 - np.maximum(np.abs(a1)[:, None, None], A3.real)
            assert np.allclose(got, want, rtol=50 * np.finfo(np.dtype(T)).eps, atol=0), np.dtype(T).name
            dest = B3.similar()
            assert dest.assign(B3 + A3) is dest and np.array_equal(dest.toarray(), A3 + A3)
        finally:
            set_plain_array_rule(old)
