"""Randomised parity: the HIP path against the CPU oracle on random strided problems -- random rank,
sizes, dim permutations, stepped / reversed / offset sub-views, broadcast (stride-0) inputs, random
fused expressions (precompiled and runtime-compiled ones), maps and reductions over random dim
subsets with random `op` / `initop`.  Deterministic seeds; arithmetic maps must agree bit for bit,
everything else within the reference's own tolerance (rtol = sqrt(eps))."""
import sys

import numpy as np
import pytest

import oraclelib
import strided_jl_amd as S
from util import fview, rtol

pytestmark = pytest.mark.gpu
fn = S.fn

# (expression, number of inputs, exact: only + - * / on reals)
EXPRS = [
    (lambda a: a, 1, True),
    (lambda a, b: a + b, 2, True),
    (lambda a, b: a * b - a, 2, True),
    (lambda a, b, c: (a + b) * c - b / 3, 3, True),
    (lambda a, b, c, d: a + b + c + d, 4, True),
    (lambda a: fn.abs2(a) + 1, 1, True),
    (lambda a, b: fn.sqrt(fn.abs(a)) * b, 2, False),
    (lambda a, b: a * fn.exp(b * 0.125) - fn.sin(a), 2, False),
]


def dview(arr):
    import torch
    a = np.asfortranarray(arr)
    t = torch.from_numpy(a.ravel(order="F").copy()).cuda()
    st, s = [], 1
    for d in a.shape:
        st.append(s)
        s *= d
    return S.StridedView(t, a.shape, tuple(st), 0)


def _random_view(rng, mk, data, dims):
    """A view of size `dims` into (a fresh copy of) `data`'s parent: random permutation of the
    parent dims, random start/step (possibly negative) per dim."""
    N = len(dims)
    perm = rng.permutation(N)
    steps = [int(rng.choice([1, 1, 1, 2, -1, 3])) for _ in range(N)]
    starts = [int(rng.integers(0, 3)) for _ in range(N)]
    pshape = [0] * N
    for i in range(N):
        pshape[perm[i]] = starts[i] + (dims[i] - 1) * abs(steps[i]) + 1 + int(rng.integers(0, 2))
    parent = data(tuple(pshape))
    V = mk(parent).permutedims(tuple(int(p) for p in perm))
    idx = []
    for i in range(N):
        n = V.size[i]
        if steps[i] > 0:
            idx.append(slice(starts[i], starts[i] + (dims[i] - 1) * steps[i] + 1, steps[i]))
        else:
            hi = starts[i] + (dims[i] - 1) * (-steps[i])
            lo = starts[i] - 1
            idx.append(slice(hi, lo if lo >= 0 else None, steps[i]))
        assert n >= starts[i] + (dims[i] - 1) * abs(steps[i]) + 1
    W = V.sview(*idx)
    assert W.size == tuple(dims), (W.size, dims)
    return W


# tools/fuzz_more.py runs the same problems on a library-owned stream, which is not ordered against torch's: it synchronises here
HOOKS = {"before": lambda: None, "after": lambda: None}


def _problem(seed, T):
    """Returns run(mk) -> result array, plus whether bit-exactness is expected."""
    rng0 = np.random.default_rng(seed)
    N = int(rng0.integers(1, 5))
    big = int(rng0.integers(0, 3)) == 0
    dims = [int(rng0.integers(1, 7)) for _ in range(N)]
    dims[int(rng0.integers(0, N))] = int(rng0.integers(20, 300 if big else 70))
    if N >= 2 and rng0.integers(0, 2):
        dims[int(rng0.integers(0, N))] = int(rng0.integers(16, 80))
    f, nin, exact = EXPRS[int(rng0.integers(0, len(EXPRS)))]
    reduce_dims = ()
    op = None
    initop = None
    if rng0.integers(0, 3) == 0:
        k = int(rng0.integers(1, N + 1))
        reduce_dims = tuple(sorted(rng0.choice(N, size=k, replace=False).tolist()))
        cplx = np.issubdtype(np.dtype(T), np.complexfloating)
        op = ["+", "+", "max", "min"][int(rng0.integers(0, 2 if cplx else 4))]
        initop = [None, "identity", "zero", ("scale", 0.5), ("const", 2.0)][int(rng0.integers(0, 5))]
        exact = False
    bmask = [[bool(rng0.integers(0, 6) == 0) for _ in range(N)] for _ in range(nin)]  # broadcast dims per input
    vseed = int(rng0.integers(0, 2 ** 31))

    def run(mk):
        rng = np.random.default_rng(vseed)  # identical draws for both backends

        def data(shape):
            x = rng.random(shape) + 0.25
            if np.issubdtype(np.dtype(T), np.complexfloating):
                x = x + 1j * (rng.random(shape) - 0.5)
            return np.asfortranarray(x.astype(T))

        ins = []
        for k in range(nin):
            d_k = [1 if bmask[k][i] else dims[i] for i in range(N)]
            ins.append(_random_view(rng, mk, data, d_k))
        odims = [1 if i in reduce_dims else dims[i] for i in range(N)]
        out = _random_view(rng, mk, data, odims)
        HOOKS["before"]()
        if op is None:
            # looked up at call time: the oracle run patches the funnel in the module
            sys.modules["strided_jl_amd.mapreduce"]._mapreduce_fuse_(f, None, None, tuple(dims), S.promoteshape(tuple(dims), out, *ins))
        else:
            S._mapreducedim_(f, op, initop, tuple(dims), (out, *ins))
        HOOKS["after"]()
        r = out.toarray()
        return r

    return run, exact, dict(N=N, dims=dims, nin=nin, reduce=reduce_dims, op=op, initop=initop)


@pytest.mark.parametrize("T", [np.float32, np.float64, np.complex128])
@pytest.mark.parametrize("chunk", range(4))
def test_random_strided_problems_match_the_oracle(chunk, T, monkeypatch):
    import torch

    def funnel(f, op, initop, dims, arrays):
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, 1)
        return arrays[0]

    for i in range(40):
        seed = 1000 * chunk + i
        run, exact, info = _problem(seed, T)
        with monkeypatch.context() as m:
            m.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
            want = run(fview)
        got = run(dview)
        torch.cuda.synchronize()
        msg = f"seed {seed} {np.dtype(T).name} {info}"
        assert got.shape == want.shape, msg
        if exact and not np.issubdtype(np.dtype(T), np.complexfloating):
            assert np.array_equal(got, want), msg
        else:
            g = got.astype(np.complex128).ravel()
            w = want.astype(np.complex128).ravel()
            assert np.linalg.norm(g - w) <= rtol(T) * max(np.linalg.norm(g), np.linalg.norm(w), 1e-300), msg
