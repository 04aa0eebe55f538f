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
