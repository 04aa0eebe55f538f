#!/usr/bin/env python3
"""Round 5 experiment: self-released launches (write-through stores, no release fence on the packet) on the EAGER path of a
library-owned stream (option eager_self_release).  The bench step issued call by call from Python; a dependent chain of 32^4
permutedims! into one destination; the same into four destinations; a read-after-write pair.  Wall clock incl. the final
synchronisation, best of 5; every result checked against torch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

n = 32
dev = torch.device("cuda", 0)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
outs = [torch.empty_like(tA) for _ in range(5)]
A = colmajor_view(S, tA, (n,) * 4)
O = [colmajor_view(S, t, (n,) * 4) for t in outs]
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
a4 = tA.reshape((n,) * 4)
ref2 = a4.permute(3, 2, 1, 0).contiguous().reshape(-1)
cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
ref3 = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)


def bench(fn, st, nsteps=1000):
    for _ in range(20):
        fn()
    st.synchronize()
    best = 1e30
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(nsteps):
            fn()
        st.synchronize()
        best = min(best, time.perf_counter() - t)
    return best / nsteps * 1e6


for mode in (0, 1):
    S.set_option("eager_self_release", mode)
    st = S.Stream()
    p2 = S.make_plan(lambda x: x, None, None, A.size, (O[0], A.permutedims((3, 2, 1, 0))))
    p3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (O[1],) + tuple(A.permutedims(p) for p in perms))
    p4 = [S.make_plan(lambda x: x, None, None, A.size, (O[k], A.permutedims((3, 2, 1, 0)))) for k in range(1, 5)]
    raw = S.make_plan(lambda x, y: x + 2 * y, None, None, A.size, (O[2], O[0], O[1]))   # reads what p2 and p3 wrote
    h = st.handle
    for t in outs:
        t.zero_()
    torch.cuda.synchronize()
    step = bench(lambda: (p2.execute(h), p3.execute(h)), st)
    chain = bench(lambda: p2.execute(h), st)
    four = bench(lambda: [p.execute(h) for p in p4], st) / 4
    p2.execute(h); p3.execute(h); raw.execute(h)
    st.synchronize()
    torch.cuda.synchronize()
    ok = torch.equal(outs[0], ref2) and torch.equal(outs[1], ref3) and torch.equal(outs[2], ref2 + 2 * ref3) and all(torch.equal(outs[k], ref2) for k in (3, 4))
    print("eager_self_release=%d | bench step %.3f us | dependent chain %.3f us per launch | four destinations %.3f us per launch | results %s"
          % (mode, step, chain, four, "bit-exact" if ok else "WRONG"), flush=True)
    st.close()
S.set_option("eager_self_release", 0)
