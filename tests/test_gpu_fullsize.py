"""BASELINE.json's full-size configurations on the GPU.  Where NumPy can check the whole result
in seconds it does (bit-exact); beyond that, size-independent properties: permute round trips,
exact integer checksums, idempotence of the symmetriser, shard additivity of the reduction."""
import numpy as np
import pytest

import strided_jl_amd as S
from strided_jl_amd import fn

pytestmark = pytest.mark.gpu


def cm(t, shape):
    st, s = [], 1
    for d in shape:
        st.append(s)
        s *= d
    return S.StridedView(t, shape, tuple(st), 0)


def test_c2_permutedims_32_4_f64_bit_exact_and_plan():
    import torch
    n = 32
    A = np.arange(n ** 4, dtype=np.float64)  # index-valued, column-major (n,n,n,n)
    tA = torch.from_numpy(A).cuda()
    tB = torch.zeros_like(tA)
    S.permutedims_(cm(tB, (n,) * 4), cm(tA, (n,) * 4), (3, 2, 1, 0))
    torch.cuda.synchronize()
    want = A.reshape((n,) * 4, order="F").transpose(3, 2, 1, 0).ravel(order="F")
    assert np.array_equal(tB.cpu().numpy(), want)
    for p in [(1, 2, 3, 0), (2, 3, 0, 1), (0, 2, 1, 3), (1, 0, 3, 2)]:
        S.permutedims_(cm(tB, (n,) * 4), cm(tA, (n,) * 4), p)
        torch.cuda.synchronize()
        assert np.array_equal(tB.cpu().numpy(), A.reshape((n,) * 4, order="F").transpose(p).ravel(order="F")), p


def test_c3_four_way_permuted_sum_32_4_f64_bit_exact():
    import torch
    n = 32
    rng = np.random.default_rng(1234)
    A = rng.standard_normal(n ** 4)
    tA = torch.from_numpy(A).cuda()
    tB = torch.zeros_like(tA)
    V = cm(tA, (n,) * 4)
    cm(tB, (n,) * 4).assign(V.permutedims((0, 1, 2, 3)) + V.permutedims((1, 2, 3, 0)) + V.permutedims((2, 3, 0, 1)) +
                            V.permutedims((3, 0, 1, 2)))
    torch.cuda.synchronize()
    a = A.reshape((n,) * 4, order="F")
    want = ((a + a.transpose(1, 2, 3, 0)) + a.transpose(2, 3, 0, 1)) + a.transpose(3, 0, 1, 2)
    assert np.array_equal(tB.cpu().numpy(), want.ravel(order="F"))


def test_c1_symmetrise_4000_f64_bit_exact_and_idempotent():
    import torch
    m = 4000
    rng = np.random.default_rng(4321)
    A = rng.standard_normal(m * m)
    tA = torch.from_numpy(A).cuda()
    tB = torch.zeros_like(tA)
    tC = torch.zeros_like(tA)
    VA, VB, VC = cm(tA, (m, m)), cm(tB, (m, m)), cm(tC, (m, m))
    VB.assign((VA + VA.adjoint()) / 2)
    torch.cuda.synchronize()
    a = A.reshape((m, m), order="F")
    assert np.array_equal(tB.cpu().numpy(), ((a + a.T) / 2).ravel(order="F"))
    VC.assign((VB + VB.adjoint()) / 2)  # symmetrising a symmetric matrix changes nothing
    torch.cuda.synchronize()
    assert torch.equal(tB, tC)


def test_permute_round_trip_and_checksum_128_4_f64():
    """2 GiB per array: B = permutedims(A, p); C = permutedims(B, inverse(p)) must be A again, and
    the integer-valued checksum of B equals that of A."""
    import torch
    n = 128
    tA = torch.randint(-1000, 1000, (n ** 4,), device="cuda", dtype=torch.int32).to(torch.float64)
    tB = torch.empty_like(tA)
    tC = torch.empty_like(tA)
    for p in [(3, 2, 1, 0), (1, 2, 3, 0), (2, 3, 0, 1)]:
        inv = tuple(int(i) for i in np.argsort(p))
        S.permutedims_(cm(tB, (n,) * 4), cm(tA, (n,) * 4), p)
        S.permutedims_(cm(tC, (n,) * 4), cm(tB, (n,) * 4), inv)
        torch.cuda.synchronize()
        assert torch.equal(tA, tC), p
        assert S.sum(cm(tB, (n,) * 4)) == S.sum(cm(tA, (n,) * 4)) == float(tA.sum().item())
        # spot-check one hyper-column against the definition
        a4 = tA.reshape((n,) * 4)  # torch (row-major) view: index order reversed
        b4 = tB.reshape((n,) * 4)
        i = (5, 17, 99, 3)
        j = [0] * 4
        for k in range(4):
            j[p[k]] = i[k]
        assert b4[i[3], i[2], i[1], i[0]] == a4[j[3], j[2], j[1], j[0]]


def test_c5_compute_bound_map_8192_f32_within_4ulp():
    import torch
    m = 8192
    tA = torch.rand(m * m, device="cuda", dtype=torch.float32)
    tB = torch.empty_like(tA)
    cm(tB, (m, m)).assign(cm(tA, (m, m)) * fn.exp(-2 * cm(tA, (m, m))) + fn.sin(cm(tA, (m, m)) * cm(tA, (m, m))))
    torch.cuda.synchronize()
    a = tA.double()
    t1, t2 = a * torch.exp(-2 * a), torch.sin(a * a)
    err = (tB.double() - (t1 + t2)).abs()
    bound = 4 * np.finfo(np.float32).eps * (t1.abs() + t2.abs()) + 1e-30
    assert bool((err <= bound).all()), float((err / bound).max())


def test_c4_abs2_sum_slab_f32_accuracy_and_shard_additivity():
    """One GPU's slab of config 4 (4096 x 4096 x 8 Float32 = 512 MiB): rtol 1e-6 against the
    float64 truth, and the sum over the 8 z-planes taken separately adds up to the whole."""
    import torch
    shape = (4096, 4096, 8)
    tA = torch.rand(int(np.prod(shape)), device="cuda", dtype=torch.float32) * 2 - 1
    V = cm(tA, shape)
    total = S.mapreduce(fn.abs2, "+", V)
    truth = float((tA.double() ** 2).sum().item())
    assert abs(total - truth) <= 1e-6 * truth
    planes = [S.mapreduce(fn.abs2, "+", V[:, :, k:k + 1]) for k in range(8)]
    assert abs(sum(planes) - truth) <= 1e-6 * truth
    # the same through dims=: one partial sum per plane in a single launch
    per_plane = S.mapreduce(fn.abs2, "+", V, dims=(0, 1)).toarray().ravel()
    assert np.allclose(per_plane, planes, rtol=1e-6)
    assert S.maximum(V, f=fn.abs) == float(tA.abs().max().item())


def test_transposes_beyond_4_gib_use_64_bit_tile_arithmetic():
    """Maximum sizes: a 5 GiB byte matrix (65536 x 81920) transposed -- tile origins and in-tile
    offsets exceed 32 bits (the WIDE kernel variant and the general-origin mode) -- and a 5 GiB Int16
    one; checked against torch's own transpose of the same buffer."""
    import torch
    for dt, rows, cols in ((torch.uint8, 65536, 81920), (torch.int16, 49152, 57344)):
        t = torch.randint(0, 200, (rows * cols,), dtype=dt, device="cuda")
        out = torch.empty_like(t)
        A = cm(t, (rows, cols))                      # column-major rows x cols
        B = cm(out, (cols, rows))
        plan = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims((1, 0))))
        d = plan.describe()
        assert "family=tiled" in d, d
        S.permutedims_(B, A, (1, 0))
        torch.cuda.synchronize()
        # column-major (rows, cols) == row-major (cols, rows): B = A^T is the row-major transpose
        want = t.view(cols, rows).t().contiguous().view(-1)
        assert torch.equal(out, want), d
        # and back: an involution
        back = torch.empty_like(t)
        S.permutedims_(cm(back, (rows, cols)), B, (1, 0))
        torch.cuda.synchronize()
        assert torch.equal(back, t)
        del t, out, back, want


def test_c4_full_size_4_gib_abs2_sum_f32():
    """configs[3] at its full single-array size (4096 x 4096 x 64 Float32 = 4 GiB) on one GPU: rtol 1e-6 against
    the float64 truth (full size was only exercised inside bench.py), and the block-partitioned
    form -- 8 slab-local sub-problems as smr_shard_ex hands them to 8 ranks -- adds up to the same value."""
    import torch
    from strided_jl_amd import distributed as D
    shape = (4096, 4096, 64)
    tA = torch.rand(int(np.prod(shape)), device="cuda", dtype=torch.float32) * 2 - 1
    V = cm(tA, shape)
    total = S.mapreduce(fn.abs2, "+", V)
    truth = 0.0
    for k in range(8):  # float64 truth slab by slab (a full float64 copy would be 8 GiB)
        truth += float((tA[k * 2 ** 27:(k + 1) * 2 ** 27].double() ** 2).sum().item())
    assert abs(total - truth) <= 1e-6 * truth
    acc = 0.0
    for r in range(8):
        slab = tA[r * 2 ** 27:(r + 1) * 2 ** 27]                  # this "rank" holds only its 512 MiB
        out = torch.zeros(1, device="cuda", dtype=torch.float32)
        O = S.StridedView(out, shape, (0, 0, 0), 0)
        A = S.StridedView(slab, shape, (1, 4096, 4096 * 4096), 0)   # logical box, slab memory
        sdims, sarr, need, sinit = D.shard(fn.abs2, "+", None, shape, (O, A), 8, r, local=(False, True))
        assert need and sdims == (4096, 4096, 8)
        S._mapreduce_fuse_(fn.abs2, "+", sinit, sdims, sarr)
        torch.cuda.synchronize()
        acc += float(out.item())
    assert abs(acc - truth) <= 1e-6 * truth


def test_orbit_family_at_64_4_and_128_4_properties():
    """The ORBIT kernel beyond what NumPy checks in seconds: the 4-way permuted sum of an INTEGER-valued array is
    exact in Float64, invariant under the cyclic permutation of its own indices (B[r i] uses the same four
    values; with integers every association gives the same sum), and equal to the classic tiled kernel's result."""
    import torch
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    for n in (64, 128):
        tA = torch.randint(-1000, 1000, (n ** 4,), device="cuda", dtype=torch.int32).to(torch.float64)
        tB = torch.empty_like(tA)
        tC = torch.empty_like(tA)
        A, B, C = cm(tA, (n,) * 4), cm(tB, (n,) * 4), cm(tC, (n,) * 4)
        plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
        assert "family=orbit" in plan.describe()
        plan.execute(int(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        b4 = tB.reshape((n,) * 4)
        assert torch.equal(b4, b4.permute(3, 0, 1, 2).contiguous())      # invariant under the rotation
        assert float(tB.sum().item()) == 4 * float(tA.sum().item())    # exact checksum
        S.set_option("orbit", 0)
        try:
            plan2 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (C,) + tuple(A.permutedims(q) for q in perms))
            assert "family=tiled" in plan2.describe()
            plan2.execute(int(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
        finally:
            S.set_option("orbit", 1)
        assert torch.equal(tB, tC)
        # in place: A .= sum of its four rotations (an orbit is read completely before it is written)
        plan3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (A,) + tuple(A.permutedims(q) for q in perms))
        if "family=orbit" in plan3.describe():
            plan3.execute(int(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            assert torch.equal(tA, tB)
        del tA, tB, tC
