#!/usr/bin/env python3
"""Are the big power-of-two problems limited by L2/HBM channel conflicts?  Same kernels on
neighbouring sizes whose strides are not powers of two (GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3


def main():
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    for n in (96, 112, 128, 144):
        tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        for name, f, arrs in (("perm4321", lambda a: a, (B, A.permutedims((3, 2, 1, 0)))),
                              ("bcast4", lambda a, b, c, d: a + b + c + d, (B,) + tuple(A.permutedims(q) for q in perms))):
            plan = S.make_plan(f, None, None, A.size, arrs)
            us = time_plan(plan, 3)
            d = plan.describe()
            print(f"{name:9s} {n}^4 f64 {us:10.1f} us {16 * n ** 4 / us / 1e3:8.1f} GB/s | {d[d.find('tile='):d.find(' algbytes')]}")
            sys.stdout.flush()
        del tA, tB
    for m in (4000, 8000, 8192, 8200, 16000, 16384):
        tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        for name, f, arrs in (("transpose", lambda a: a, (B, A.permutedims((1, 0)))), ("sym", lambda x, y: (x + y) / 2, (B, A, A.permutedims((1, 0))))):
            plan = S.make_plan(f, None, None, A.size, arrs)
            us = time_plan(plan, 5)
            d = plan.describe()
            print(f"{name:9s} {m}^2 f64 {us:10.1f} us {16 * m * m / us / 1e3:8.1f} GB/s | {d[d.find('tile='):d.find(' algbytes')]}")
            sys.stdout.flush()
        del tA, tB


if __name__ == "__main__":
    main()
