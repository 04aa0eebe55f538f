"""Per-operation typing of fused expressions: Julia carries out `Float32 .* Float32` in Float32 even
when the destination (or another operand) is Float64; the device computes a call in one class, so
the host inserts ROUND32 after every operation whose Julia type is Float32 / ComplexF32.  With that
the real arithmetic (+ - * / sqrt) of mixed-precision expressions is bit-identical to NumPy's
per-operation typing.  CPU: the oracle; GPU: the HIP kernels (interpreter and runtime-compiled)."""
import sys

import numpy as np
import pytest

import oraclelib
import strided_jl_amd as S
from util import fview

fn = S.fn


def _check(mk, sync=lambda: None):
    rng = np.random.default_rng(77)
    a = rng.standard_normal((70, 50)).astype(np.float32)
    b = np.abs(rng.standard_normal((70, 50))).astype(np.float32) + np.float32(0.5)
    d = rng.standard_normal((70, 50))                       # Float64 operand
    A, B, Dv = mk(a), mk(b), mk(d)
    out = mk(np.zeros((70, 50)))
    # Float32 product, widened, then a Float64 constant
    out.assign(A * B - 0.5)
    sync()
    assert np.array_equal(out.toarray(), (a * b).astype(np.float64) - 0.5)
    # weak integer constants keep Float32; the division too; sqrt of a Float32 stays Float32
    out.assign(fn.sqrt(B) * 3 + A / 7)
    sync()
    assert np.array_equal(out.toarray(), (np.sqrt(b) * np.float32(3) + a / np.float32(7)).astype(np.float64))
    # mixed operands: (Float32 + Float32) is rounded, the product with a Float64 array is not
    out.assign((A + B) * Dv)
    sync()
    assert np.array_equal(out.toarray(), (a + b).astype(np.float64) * d)
    # transposed Float32 input into a Float64 destination (tiled family)
    outT = mk(np.zeros((50, 70)))
    outT.assign(A.permutedims((1, 0)) * B.permutedims((1, 0)) + 1.25)
    sync()
    assert np.array_equal(outT.toarray(), (a.T * b.T).astype(np.float64) + 1.25)
    # a Float64 SCALAR among Float32 arrays: Julia multiplies in Float64 and rounds once on store (SMR_OP_WIDEN)
    o32 = mk(np.zeros((70, 50), dtype=np.float32))
    o32.assign(A * 0.1)
    sync()
    assert np.array_equal(o32.toarray(), (a.astype(np.float64) * 0.1).astype(np.float32))
    assert not np.array_equal(o32.toarray(), a * np.float32(0.1))  # the Float32 product differs in the last bit somewhere
    o32.assign(A * B + 0.3)   # Float32 product (rounded), Float64 sum, one rounding on store
    sync()
    assert np.array_equal(o32.toarray(), ((a * b).astype(np.float64) + 0.3).astype(np.float32))
    o32.assign(A * np.float32(0.1) + 2)   # a Float32 scalar and a weak integer: everything stays Float32
    sync()
    assert np.array_equal(o32.toarray(), a * np.float32(0.1) + np.float32(2))
    # min / max follow Julia: NaN if either argument is NaN, min(-0.0, 0.0) = -0.0, max(-0.0, 0.0) = 0.0
    x = np.array([[np.nan, 1.0, -0.0, 0.0, 3.0, np.nan]] * 2, dtype=np.float64)
    y = np.array([[2.0, np.nan, 0.0, -0.0, -1.0, np.nan]] * 2, dtype=np.float64)
    X, Y, o = mk(x), mk(y), mk(np.zeros_like(x))
    o.assign(fn.min(X, Y))
    sync()
    got = o.toarray()[0]
    assert np.isnan(got[0]) and np.isnan(got[1]) and np.isnan(got[5]) and got[4] == -1.0
    assert np.signbit(got[2]) and np.signbit(got[3]) and got[2] == 0 and got[3] == 0
    o.assign(fn.max(X, Y))
    sync()
    got = o.toarray()[0]
    assert np.isnan(got[0]) and np.isnan(got[1]) and np.isnan(got[5]) and got[4] == 3.0
    assert not np.signbit(got[2]) and not np.signbit(got[3])
    # ... and so do minimum / maximum: one NaN anywhere gives NaN (the reference seeds with first(A),
    # src/mapreduce.jl:62-66,184-185; seeding with +-inf and Julia's min / max gives the same value for every input)
    z = rng.standard_normal((33, 21))
    z[17, 5] = np.nan
    assert np.isnan(S.minimum(mk(z))) and np.isnan(S.maximum(mk(z)))
    zz = np.where(np.isnan(z), 0.5, z)
    assert S.minimum(mk(zz)) == zz.min() and S.maximum(mk(zz)) == zz.max()
    assert np.signbit(S.minimum(mk(np.array([[0.0, -0.0], [0.0, 0.0]])))) and not np.signbit(S.maximum(mk(np.array([[-0.0, 0.0], [-0.0, -0.0]]))))
    # reductions: the mapped value is a Float32 product, accumulated in the destination's Float64
    acc = mk(np.zeros((70, 1)))
    S.mapreducedim_(lambda x, y: x * y, "+", acc, A, B)
    sync()
    want = (a * b).astype(np.float64).sum(axis=1, keepdims=True)
    assert np.allclose(acc.toarray(), want, rtol=1e-13, atol=0)


def test_oracle_per_operation_typing(monkeypatch):
    def funnel(f, op, initop, dims, arrays):
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, 1)
        return arrays[0]

    monkeypatch.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
    _check(fview)


def test_round32_is_only_emitted_when_the_call_computes_wider():
    a32 = S.StridedView(np.zeros((8, 8), dtype=np.float32, order="F"))
    o32, o64 = a32.similar(), S.StridedView(np.zeros((8, 8), order="F"))
    p, _ = S.build_problem(lambda x: x * x - 1, None, None, a32.size, (o32, a32), stream=0)
    assert S._lib.OPCODES["ROUND32"] not in bytes(p.fprog[0:2 * p.fprog_len])[0::2]
    p, _ = S.build_problem(lambda x: x * x - 1, None, None, a32.size, (o64, a32), stream=0)
    ops = list(bytes(p.fprog[0:2 * p.fprog_len])[0::2])
    assert ops.count(S._lib.OPCODES["ROUND32"]) == 2      # after the product and after the (weak) subtraction


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [1, 0])
def test_hip_per_operation_typing(jit):
    import torch
    from test_gpu_parity import dview
    S.set_option("jit", jit)
    try:
        _check(dview, torch.cuda.synchronize)
    finally:
        S.set_option("jit", 1)
