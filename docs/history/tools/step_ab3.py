#!/usr/bin/env python3
"""What the kernel BEFORE the 4-way sum has to do for the sum to slow down: only read other data, only write other data."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

fn = S.fn


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def t(fn_, reps=400):
    fn_()
    torch.cuda.synchronize()
    g = graph_of(torch, fn_, reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(7)) / reps * 1e3


n = 32
tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
tC = torch.empty_like(tA)
A, C = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tC, (n,) * 4)
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
sum4 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in perms))
others = {}
for mib in (2, 8, 32):
    m = mib * (1 << 20) // 8
    tX = torch.randn(m, dtype=torch.float64, device="cuda")
    tY = torch.empty_like(tX)
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    X, Y = S.StridedView(tX, (m,), (1,), 0), S.StridedView(tY, (m,), (1,), 0)
    O = S.StridedView(out, (m,), (0,), 0)
    others[f"read-only {mib} MiB (sum of X)"] = (S.make_plan(fn.abs2, "+", "zero", (m,), (O, X)), (tX, out))
    for nts in (0, 1):
        S.set_option("nt_store", nts)
        others[f"write-only {mib} MiB (Y .= 1.5), nt={nts}"] = (S.make_plan(lambda: 1.5, None, None, (m,), (Y,)), (tY,))
    S.set_option("nt_store", -1)
    others[f"copy X -> Y {mib}+{mib} MiB"] = (S.make_plan(lambda x: x, None, None, (m,), (Y, X)), (tX, tY))
a4 = t(lambda: sum4.execute(cur()))
print(f"sum4 alone {a4:.2f} us")
for name, (p, keep) in others.items():
    al = t(lambda: p.execute(cur()))

    def pair():
        sum4.execute(cur())
        p.execute(cur())
    tp = t(pair)
    print(f"{name:42s} alone {al:6.2f} | pair {tp:6.2f} | penalty {tp - al - a4:+.2f}")
