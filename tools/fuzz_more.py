#!/usr/bin/env python3
"""More seeds of tests/test_gpu_fuzz.py's randomised GPU-vs-oracle comparison (GPU box only).
Usage: [BIG=1] [OWN_STREAM=1] python tools/fuzz_more.py [first_seed] [count]
BIG: two or three long dims (a few million elements); OWN_STREAM: every call runs on a library-owned stream (eager direct dispatch,
self-released launches) instead of HIP's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oraclelib  # noqa: E402
import strided_jl_amd as S  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from util import fview, rtol  # noqa: E402


# a library-owned stream is not ordered against torch's streams (S.Stream's contract): operands torch uploaded must be complete before
# the call, the stream synchronised before torch reads the result
HOOKS = {"before": lambda: None, "after": lambda: None}


def problem_big(seed, T):
    """Like tests/test_gpu_fuzz.py:_problem but with two or three long dims (tiled / ROW / COL / split
    paths with ragged extents), total size up to a few million elements."""
    rng0 = np.random.default_rng(seed)
    N = int(rng0.integers(2, 5))
    dims = [int(rng0.integers(1, 6)) for _ in range(N)]
    longs = rng0.choice(N, size=min(N, int(rng0.integers(2, 4))), replace=False)
    budget = 3_000_000
    for i in longs:
        cap = max(8, int(min(700, budget // max(1, int(np.prod(dims))))))
        dims[int(i)] = int(rng0.integers(min(40, cap - 1), cap))
    f, nin, exact = F.EXPRS[int(rng0.integers(0, len(F.EXPRS)))]
    reduce_dims, op, initop = (), None, None
    if rng0.integers(0, 3) == 0:
        k = int(rng0.integers(1, N))
        reduce_dims = tuple(sorted(rng0.choice(N, size=k, replace=False).tolist()))
        cplx = np.issubdtype(np.dtype(T), np.complexfloating)
        op = ["+", "+", "max", "min"][int(rng0.integers(0, 2 if cplx else 4))]
        initop = [None, "identity", "zero", ("scale", 0.5), ("const", 2.0)][int(rng0.integers(0, 5))]
        exact = False
    bmask = [[bool(rng0.integers(0, 8) == 0) for _ in range(N)] for _ in range(nin)]
    alias = bool(rng0.integers(0, 2))  # inputs as views of ONE parent (orbit order) when shapes allow
    vseed = int(rng0.integers(0, 2 ** 31))

    def run(mk):
        rng = np.random.default_rng(vseed)

        def data(shape):
            x = rng.random(shape) + 0.25
            if np.issubdtype(np.dtype(T), np.complexfloating):
                x = x + 1j * (rng.random(shape) - 0.5)
            return np.asfortranarray(x.astype(T))

        ins = []
        shared = None
        for k in range(nin):
            d_k = [1 if bmask[k][i] else dims[i] for i in range(N)]
            if alias and len(set(dims)) == 1 and not any(bmask[k]):
                if shared is None:
                    shared = mk(data(tuple(dims)))
                ins.append(shared.permutedims(tuple(int(q) for q in rng.permutation(N))))
            else:
                ins.append(F._random_view(rng, mk, data, d_k))
        odims = [1 if i in reduce_dims else dims[i] for i in range(N)]
        out = F._random_view(rng, mk, data, odims)
        mod = sys.modules["strided_jl_amd.mapreduce"]
        HOOKS["before"]()
        if op is None:
            mod._mapreduce_fuse_(f, None, None, tuple(dims), S.promoteshape(tuple(dims), out, *ins))
        else:
            S._mapreducedim_(f, op, initop, tuple(dims), (out, *ins))
        HOOKS["after"]()
        return out.toarray()

    return run, exact, dict(N=N, dims=dims, nin=nin, reduce=reduce_dims, op=op, initop=initop, alias=alias)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    mod = sys.modules["strided_jl_amd.mapreduce"]
    real = mod._mapreduce_fuse_

    def funnel(f, op, initop, dims, arrays):
        p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
        oraclelib.mapreduce(p, 1)
        return arrays[0]

    bad = 0
    own = S.Stream() if os.environ.get("OWN_STREAM") else None
    if own is not None:
        HOOKS["before"] = F.HOOKS["before"] = torch.cuda.synchronize
        HOOKS["after"] = F.HOOKS["after"] = own.synchronize
    for seed in range(first, first + count):
        for T in (np.float32, np.float64, np.complex64, np.complex128):
            run, exact, info = (problem_big if os.environ.get("BIG") else F._problem)(seed, T)
            mod._mapreduce_fuse_ = funnel
            try:
                want = run(fview)
            finally:
                mod._mapreduce_fuse_ = real
            if own is not None:
                with own:
                    got = run(F.dview)
            else:
                got = run(F.dview)
            torch.cuda.synchronize()
            ok = got.shape == want.shape
            if ok:
                if exact and not np.issubdtype(np.dtype(T), np.complexfloating):
                    ok = np.array_equal(got, want)
                else:
                    g, w = got.astype(np.complex128).ravel(), want.astype(np.complex128).ravel()
                    # the oracle accumulates serially in the element type like the reference (error ~ n eps):
                    # allow for that on long float32 reductions
                    nred = int(np.prod([info["dims"][i] for i in info["reduce"]])) if info["reduce"] else 1
                    tol = rtol(T) + nred * float(np.finfo(np.dtype(T)).eps) / 8
                    ok = np.linalg.norm(g - w) <= tol * max(np.linalg.norm(g), np.linalg.norm(w), 1e-300)
            if not ok:
                bad += 1
                print("MISMATCH", seed, np.dtype(T).name, info, flush=True)
    print(f"seeds {first}..{first + count - 1} x 4 dtypes: {bad} mismatches")


if __name__ == "__main__":
    main()
