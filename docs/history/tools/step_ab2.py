#!/usr/bin/env python3
"""Which property of the 4-way sum makes it pay ~0.9 us when it alternates with another kernel in a graph?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def t(fn, reps=500):
    fn()
    torch.cuda.synchronize()
    g = graph_of(torch, fn, reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(7)) / reps * 1e3


n = 32
tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
tB, tC, tD = (torch.empty_like(tA) for _ in range(3))
A, B, C, D = (colmajor_view(S, x, (n,) * 4) for x in (tA, tB, tC, tD))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
views = tuple(A.permutedims(p) for p in perms)
add4 = lambda a, b, c, d: a + b + c + d  # noqa: E731
pc = S.make_plan(lambda x: x, None, None, A.size, (D, A))
variants = {}
variants["orbit"] = S.make_plan(add4, None, None, A.size, (C,) + views)
S.set_option("nt_store", 1)
variants["orbit nt"] = S.make_plan(add4, None, None, A.size, (C,) + views)
S.set_option("nt_store", -1)
S.set_option("orbit", 0)
variants["classic tiled"] = S.make_plan(add4, None, None, A.size, (C,) + views)
S.set_option("orbit", 1)
S.set_option("orbit_lg", 3)
variants["orbit 8^4"] = S.make_plan(add4, None, None, A.size, (C,) + views)
S.set_option("orbit_lg", -1)
m = 1024
tM = torch.randn(m * m, dtype=torch.float64, device="cuda")
tN = torch.empty_like(tM)
M, N = colmajor_view(S, tM, (m, m)), colmajor_view(S, tN, (m, m))
variants["orbit sym 1024^2"] = S.make_plan(lambda x, y: (x + y) / 2, None, None, (m, m), (N, M, M.adjoint()))
ca = t(lambda: pc.execute(cur()))
print(f"copy alone {ca:.2f}")
for name, p in variants.items():
    al = t(lambda: p.execute(cur()))

    def pair():
        p.execute(cur())
        pc.execute(cur())

    def quad():
        p.execute(cur())
        p.execute(cur())
        p.execute(cur())
        pc.execute(cur())
    tp, tq = t(pair), t(quad, 250)
    d = p.describe()
    print(f"{name:18s} alone {al:5.2f} | +copy pair {tp:5.2f} (penalty {tp - al - ca:+.2f}) | x3 +copy {tq:5.2f} (penalty {tq - 3 * al - ca:+.2f}) | {d[d.find('family='):d.find(' ct=')]}")
