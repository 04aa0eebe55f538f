"""Known-answer tests of the oracle's restatement of the reference planner:
_mapreduce_fuse! / _mapreduce_order! / _computeblocks / _mapreduce_threaded!
(/root/reference/src/mapreduce.jl:98-227, 427-520).  Expected values: SURVEY.md Appendix B
(hand-traced from the cited code)."""
import numpy as np
import pytest

import oraclelib
import strided_jl_amd as S


def _problem(dims, strides, dtype, f=None, op=None):
    """strides[0] = destination.  Builds views over one big scratch parent each."""
    views = []
    for st in strides:
        span = 1 + sum((d - 1) * abs(s) for d, s in zip(dims, st))
        parent = np.zeros(span, dtype=dtype)
        views.append(S.StridedView(parent, dims, st, 0, "identity"))
    nin = len(strides) - 1
    if f is None:
        f = {1: lambda a: a, 2: lambda a, b: a + b, 4: lambda a, b, c, d: a + b + c + d}[nin]
    p, keep = S.build_problem(f, op, None, dims, tuple(views), stream=0)
    return p, keep


def test_indexorder():
    # src/mapreduce.jl:427-441: rank of |stride| among non-zero strides; zero -> 1; ties equal
    assert oraclelib.indexorder((1, 32, 1024, 32768)) == (1, 2, 3, 4)
    assert oraclelib.indexorder((32768, 1024, 32, 1)) == (4, 3, 2, 1)
    assert oraclelib.indexorder((0, 4, 0, 2)) == (1, 2, 1, 1)
    assert oraclelib.indexorder((-8, 2, 2)) == (3, 1, 1)


def test_c1_symmetrise_4000():
    p, k = _problem((4000, 4000), [(1, 4000), (1, 4000), (4000, 1)], np.float64)
    pl = oraclelib.plan(p)
    assert pl["g"] == 3
    assert pl["fused"] == (4000, 4000)
    assert pl["importance"] == (25, 11)
    assert pl["perm"] == (0, 1)
    assert pl["costs"] == (2, 2)
    assert pl["blocks"] == (40, 32)


def test_c2_permutedims_32_4():
    p, k = _problem((32,) * 4, [(1, 32, 1024, 32768), (32768, 1024, 32, 1)], np.float64)
    pl = oraclelib.plan(p)
    assert pl["g"] == 2
    assert pl["importance"] == (129, 36, 24, 66)
    assert pl["perm"] == (0, 3, 1, 2)
    assert pl["strides"][0] == (1, 32768, 32, 1024)
    assert pl["strides"][1] == (32768, 1, 1024, 32)
    assert pl["costs"] == (2, 2, 64, 64)
    assert pl["blocks"] == (32, 32, 2, 1)


def test_c3_four_way_sum_32_4():
    st = [(1, 32, 1024, 32768), (1, 32, 1024, 32768), (32, 1024, 32768, 1), (1024, 32768, 1, 32), (32768, 1, 32, 1024)]
    p, k = _problem((32,) * 4, st, np.float64)
    pl = oraclelib.plan(p)
    assert pl["g"] == 3
    assert pl["importance"] == (1609, 713, 601, 587)
    assert pl["perm"] == (0, 1, 2, 3)
    assert pl["costs"] == (2, 2, 2, 2)
    assert pl["blocks"] == (6, 5, 4, 4)


def test_c4_complete_reduction_fuses_to_1d():
    dims = (4096, 4096, 64)
    p, k = _problem(dims, [(0, 0, 0), (1, 4096, 4096 * 4096)], np.float32, f=S.fn.abs2, op="+")
    pl = oraclelib.plan(p)
    assert pl["fused"] == (2 ** 30, 1, 1)
    assert pl["g"] == 2
    assert pl["importance"] == (48, 0, 0)
    assert pl["perm"] == (0, 1, 2)
    assert pl["costs"] == (1, 1, 1)
    assert pl["blocks"] == (2 ** 30, 1, 1)


def test_c5_elementwise_8192():
    p, k = _problem((8192, 8192), [(1, 8192)] * 5, np.float32)
    pl = oraclelib.plan(p)
    assert pl["fused"] == (2 ** 26, 1)
    assert pl["perm"] == (0, 1)
    assert pl["costs"] == (2, 16384)
    assert pl["blocks"] == (2 ** 26, 1)


def test_scaled_transpose_1000():
    p, k = _problem((1000, 1000), [(1, 1000), (1000, 1)], np.float64)
    pl = oraclelib.plan(p)
    assert pl["g"] == 2
    assert pl["importance"] == (9, 6)
    assert pl["blocks"] == (43, 42)


@pytest.mark.parametrize("srcstr,fused,imp,perm,dims,costs,blocks", [
    # permutedims!(B, A, (2,3,4,1)), benchmarks/benchtests.jl:41
    ((32, 1024, 32768, 1), (32768, 1, 1, 32), (144, 0, 0, 66), (0, 3, 1, 2), (32768, 32, 1, 1),
     (2, 2, 64, 2048), (64, 32, 1, 1)),
    # permutedims!(B, A, (3,4,1,2)), benchmarks/benchtests.jl:42
    ((1024, 32768, 1, 32), (1024, 1, 1024, 1), (132, 0, 72, 0), (0, 2, 1, 3), (1024, 1024, 1, 1),
     (2, 2, 64, 64), (64, 32, 1, 1)),
])
def test_bench_permutations(srcstr, fused, imp, perm, dims, costs, blocks):
    p, k = _problem((32,) * 4, [(1, 32, 1024, 32768), srcstr], np.float64)
    pl = oraclelib.plan(p)
    assert pl["fused"] == fused
    assert pl["importance"] == imp
    assert pl["perm"] == perm
    assert pl["dims"] == dims
    assert pl["costs"] == costs
    assert pl["blocks"] == blocks


def test_generic_matmul_103_complex_int_sized():
    # __mul! on 103x103 with 16-byte elements (Complex{Int} in the reference; c128 has the same size)
    m = n = kk = 103
    st = [(1, 103, 0), (1, 0, 103), (0, 103, 1)]
    p, k = _problem((m, n, kk), st, np.complex128, f=lambda x, y: x * y, op="+")
    pl = oraclelib.plan(p)
    assert pl["g"] == 3
    assert pl["importance"] == (256, 88, 200)
    assert pl["perm"] == (0, 2, 1)
    assert pl["strides"][0] == (1, 0, 103)
    assert pl["costs"] == (1, 1, 1)
    assert pl["blocks"] == (103, 40, 40)


def test_threaded_bisection_c1_4_threads():
    # src/mapreduce.jl:203-222: i = _lastargmax((dims .- 1) .* costs).  Level 1: (7998, 7998) ->
    # tie -> LAST index -> dim 2 halves to 2000; level 2: (7998, 3998) -> dim 1 halves to 2000.
    # (SURVEY.md App. B says "dim 2 twice"; tracing the cited code gives 2000 x 2000 boxes.)
    p, k = _problem((4000, 4000), [(1, 4000), (1, 4000), (4000, 1)], np.float64)
    boxes = oraclelib.threaded_boxes(p, 4)
    assert [b[0] for b in boxes] == [(2000, 2000)] * 4
    assert sorted(b[1][0] for b in boxes) == [0, 2000, 4000 * 2000, 4000 * 2000 + 2000]


def test_threaded_bisection_c2_4_threads():
    p, k = _problem((32,) * 4, [(1, 32, 1024, 32768), (32768, 1024, 32, 1)], np.float64)
    boxes = oraclelib.threaded_boxes(p, 4)
    assert [b[0] for b in boxes] == [(32, 32, 16, 16)] * 4


def test_threaded_never_splits_reduction_dims():
    # partial reduction: output stride 0 on dim 2 -> cost zeroed -> only dims 1/3 may be split
    dims = (64, 4096, 64)
    p, k = _problem(dims, [(1, 0, 64), (1, 64, 64 * 4096)], np.float64, f=lambda a: a, op="+")
    boxes = oraclelib.threaded_boxes(p, 8)
    assert len(boxes) > 1
    pl = oraclelib.plan(p)
    red = [i for i, s in enumerate(pl["strides"][0]) if s == 0]
    for b, _ in boxes:
        for i in red:
            assert b[i] == pl["dims"][i]


def test_small_problems_are_not_threaded():
    p, k = _problem((32, 32, 32), [(1, 32, 1024), (1024, 32, 1)], np.float64)  # 32768 = MINTHREADLENGTH
    assert len(oraclelib.threaded_boxes(p, 8)) == 1
