"""Pins the CPU oracle: every case of the reference's test list (tests/cases.py, restating
/root/reference/test/othertests.jl) is run through the public front-ends with the device funnel
replaced by the oracle, single- and multi-threaded like test/runtests.jl:12-20, and compared with
the NumPy restatement of the Base-Julia side of each assertion."""
import numpy as np
import pytest

import cases
import oraclelib
import strided_jl_amd as S
from util import fview, rtol


@pytest.fixture(params=[1, 4], ids=["1thread", "4threads"])
def oracle_engine(request, monkeypatch):
    nthreads = request.param

    def funnel(f, op, initop, dims, arrays):
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, nthreads)
        return arrays[0]

    import sys
    monkeypatch.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
    return nthreads


def _check(res, exp, exact, name):
    assert len(res) == len(exp)
    for i, (r, e) in enumerate(zip(res, exp)):
        r, e = np.asarray(r), np.asarray(e)
        assert r.shape == e.shape, f"{name}[{i}]: shape {r.shape} vs {e.shape}"
        if exact:
            assert np.array_equal(r, e), f"{name}[{i}] not bit-exact"
        else:
            # rtol = sqrt(eps) of the coarser side: with mixed precisions (Float32 arrays, Float64
            # literal) Julia types every operation separately while the engine evaluates the whole
            # fused expression in the widest class, so they agree to the narrower type's eps only
            tol = max(rtol(x.dtype if np.issubdtype(x.dtype, np.inexact) else np.float64) for x in (r, e))
            # Julia isapprox on arrays: norm(x - y) <= rtol * max(norm(x), norm(y))
            num = np.linalg.norm((r.astype(np.complex128) - e.astype(np.complex128)).ravel())
            den = max(np.linalg.norm(r.astype(np.complex128).ravel()), np.linalg.norm(e.astype(np.complex128).ravel()))
            assert num <= tol * den, f"{name}[{i}]: rel err {num / max(den, 1e-300):.3e} > {tol:.1e}"


@pytest.mark.parametrize("name", cases.NAMES)
def test_oracle_matches_numpy(name, oracle_engine):
    res, exp, exact = cases.run_case(name, fview)
    _check(res, exp, exact, name)


def _reversed_dest_problem(mk):
    """out[k] = 2 + sum over (i, j) of (x + y + z)[i, j, k], out seen through a REVERSED range (destination stride -1 along the kept
    dim), inputs with reversed ranges too (their negative strides make every block weight <= 0, oracle :461, and the kept dim is cut
    into blocks of 40).  Layout of tools/fuzz_more.py's BIG seed 63092, which found this in round 5."""
    rng = np.random.default_rng(63092)
    dims = (298, 20, 159)
    layout = [((483, 0, 1), 323, 144417), ((-3220, -161, 1), 962621, 966000), ((3, 143040, 894), 143041, 3146880)]
    ins, total = [], 0.0
    for strides, offset, n in layout:
        host = rng.random(n) + 0.25
        ins.append(S.StridedView(mk(host).parent, dims, strides, offset))
        total = total + S.StridedView(host, dims, strides, offset).toarray()
    stale = rng.random(1431) + 0.25                     # what the destination's parent holds before the call
    parent = mk(stale.copy())
    out = S.StridedView(parent.parent, dims, (0, 0, -1), 1430)
    want = stale.copy()
    want[1430 - np.arange(159)] = 2.0 + total.sum(axis=(0, 1))
    return tuple(ins), out, parent, want, stale


def test_initop_with_a_reversed_destination():
    """src/mapreduce.jl:409 keeps `init` alive across the blocks of a dim only while the destination's stride is > 0.  Read literally,
    a reversed destination loses its initop in every block after the first: the result depends on the old contents and on the block
    size.  The oracle reads the line as `!= 0` (header of oracle/strided_oracle.cpp); this test shows both readings against NumPy."""
    ins, out, parent, want, stale = _reversed_dest_problem(fview)
    p, keep = S.build_problem(lambda x, y, z: x + y + z, "+", ("const", 2.0), out.size, (out,) + ins, stream=0)
    oraclelib.mapreduce(p, 1)
    got = parent.toarray().ravel()
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    info = oraclelib.plan(p)
    kept = [i for i in range(info["N"]) if info["strides"][0][i] != 0]
    assert len(kept) == 1 and info["blocks"][kept[0]] < info["dims"][kept[0]]   # the kept dim really is cut into several blocks
    try:
        oraclelib.set_literal_409(True)
        parent.parent[...] = stale
        oraclelib.mapreduce(p, 1)
        lit = parent.toarray().ravel()
    finally:
        oraclelib.set_literal_409(False)
    nblk = int(info["blocks"][kept[0]])
    bad = np.nonzero(~np.isclose(lit, want, rtol=1e-12, atol=0))[0]
    assert len(bad) == 159 - nblk == 119                 # every output outside the first block of 40 ...
    assert np.allclose(lit[bad] - want[bad], stale[bad] - 2.0)   # ... was accumulated onto its stale value instead of the constant
    assert sorted(bad.tolist()) == list(range(1430 - 158, 1430 - 39))


@pytest.mark.gpu
def test_hip_initop_with_a_reversed_destination():
    import torch
    from test_gpu_fuzz import dview
    ins, out, parent, want, stale = _reversed_dest_problem(dview)
    S._mapreducedim_(lambda x, y, z: x + y + z, "+", ("const", 2.0), out.size, (out,) + ins)
    torch.cuda.synchronize()
    assert np.allclose(parent.toarray().ravel(), want, rtol=1e-11, atol=0)   # (sums of 5960 terms in another order)
