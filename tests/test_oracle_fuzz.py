"""Randomised pinning of the CPU oracle against NumPy: the same random strided problems as
tests/test_gpu_fuzz.py (random rank / sizes / permutations / stepped, reversed and offset sub-views /
broadcast inputs / fused expressions / reductions with every `initop` form), with the expected
result computed by plain NumPy on the materialised operands.  Runs single- and multi-threaded (the
reference's task bisection)."""
import sys

import numpy as np
import pytest

import oraclelib
import strided_jl_amd as S
import test_gpu_fuzz as F
from util import fview, rtol

NP_EXPRS = [
    lambda a: a,
    lambda a, b: a + b,
    lambda a, b: a * b - a,
    lambda a, b, c: (a + b) * c - b / 3,
    lambda a, b, c, d: a + b + c + d,
    lambda a: (a.real * a.real + a.imag * a.imag if np.iscomplexobj(a) else a * a) + 1,
    lambda a, b: np.sqrt(np.abs(a)) * b,
    lambda a, b: a * np.exp(b * 0.125) - np.sin(a),
]


@pytest.mark.parametrize("nthreads", [1, 3])
@pytest.mark.parametrize("T", [np.float32, np.float64, np.complex128])
def test_oracle_matches_numpy_on_random_strided_problems(T, nthreads, monkeypatch):
    captured = {}

    def funnel(f, op, initop, dims, arrays):
        # NumPy truth from the operands as they are BEFORE the engine runs
        ins = [np.broadcast_to(a.toarray(), dims) for a in arrays[1:]]
        old = arrays[0].toarray()
        # the destination arrives promoted (stride 0 along reduced dims): keep one copy of each element
        old = old[tuple(slice(0, 1) if st == 0 else slice(None) for st in arrays[0].strides)]
        captured["truth_inputs"] = (ins, old)
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, nthreads)
        return arrays[0]

    monkeypatch.setattr(sys.modules["strided_jl_amd.mapreduce"], "_mapreduce_fuse_", funnel)
    for seed in list(range(0, 60)) + list(range(1000, 1060)):
        run, exact, info = F._problem(seed, T)
        got = run(fview)
        ins, old = captured["truth_inputs"]
        idx = _expr_index(seed)
        wide = np.complex128 if np.issubdtype(np.dtype(T), np.complexfloating) else np.float64
        val = NP_EXPRS[idx](*[x.astype(wide) for x in ins])
        if info["op"] is None:
            want = val
        else:
            red = {"+": np.sum, "max": np.max, "min": np.min}[info["op"]](val.real if info["op"] != "+" and np.iscomplexobj(val) else val,
                                                                       axis=info["reduce"], keepdims=True)
            o = old.astype(wide)
            io = info["initop"]
            start = {None: o, "identity": o, "zero": o * 0}.get(io if not isinstance(io, tuple) else None, None)
            if isinstance(io, tuple):
                start = o * io[1] if io[0] == "scale" else np.full_like(o, io[1])
            comb = {"+": np.add, "max": np.maximum, "min": np.minimum}[info["op"]]
            want = comb(start, red)
        g = got.astype(np.complex128).ravel()
        w = np.asarray(want).astype(np.complex128).ravel()
        assert g.shape == w.shape, (seed, info)
        assert np.linalg.norm(g - w) <= rtol(T) * max(np.linalg.norm(w), 1e-300), (seed, info)


def _expr_index(seed):
    """Replays tests/test_gpu_fuzz.py:_problem's random draws up to the expression choice."""
    rng0 = np.random.default_rng(seed)
    N = int(rng0.integers(1, 5))
    big = int(rng0.integers(0, 3)) == 0
    [int(rng0.integers(1, 7)) for _ in range(N)]
    int(rng0.integers(0, N))
    int(rng0.integers(20, 300 if big else 70))
    if N >= 2 and rng0.integers(0, 2):
        int(rng0.integers(0, N))
        int(rng0.integers(16, 80))
    return int(rng0.integers(0, len(F.EXPRS)))
