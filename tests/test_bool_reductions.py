"""`&` / `|` reductions (`mapreduce(pred, &, A)`, `mapreduce(pred, |, A; dims)`): neutral elements
true / false as in `_init_reduction!` (src/mapreduce.jl:188-189).  CPU: oracle vs NumPy (single- and
multi-threaded complete reduction = per-task partial slots); GPU: HIP vs NumPy, all three reduction
kernels (complete, ROW/COL partial, general partial)."""
import sys

import numpy as np
import pytest

import oraclelib
import strided_jl_amd as S
from util import fview

fn = S.fn


def _check(mk, shape=(37, 20, 11)):
    rng = np.random.default_rng(31)
    a = rng.standard_normal(shape)
    A = mk(a)
    P = A.permutedims((2, 0, 1))
    assert bool(S.mapreduce(lambda x: x > -10, "&", A)) is True
    assert bool(S.mapreduce(lambda x: x > 0, "&", A)) is False
    assert bool(S.mapreduce(lambda x: x > 10, "|", P)) is False
    assert bool(S.mapreduce(lambda x: x > 2.5, "|", P)) == bool((a > 2.5).any())
    for dims in ((0,), (1, 2), (2,)):
        got = S.mapreduce(lambda x: x > -1.0, "&", A, dims=dims).toarray()
        assert got.dtype == np.bool_ and np.array_equal(got, (a > -1.0).all(axis=dims, keepdims=True))
        got = S.mapreduce(lambda x: x > 1.5, "|", A, dims=dims).toarray()
        assert np.array_equal(got, (a > 1.5).any(axis=dims, keepdims=True))
    # accumulate into existing destination content: out .= out & all(...)
    out = mk(np.array([[[True]], [[False]]] * 1).reshape(2, 1, 1))
    B = mk(np.ones((2, 5, 7)))
    S.mapreducedim_(lambda x: x > 0, "&", out, B)
    assert out.toarray().ravel().tolist() == [True, False]


@pytest.mark.parametrize("nthreads", [1, 4])
def test_oracle_and_or_reductions(nthreads, monkeypatch):
    def funnel(f, op, initop, dims, arrays):
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, nthreads)
        return arrays[0]

    monkeypatch.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
    _check(fview)
    _check(fview, (70, 33, 41))  # > 32768 elements: the threaded branches of the oracle


@pytest.mark.gpu
def test_hip_and_or_reductions():
    import torch
    from test_gpu_parity import dview
    _check(dview)
    _check(dview, (300, 64, 40))
    torch.cuda.synchronize()
