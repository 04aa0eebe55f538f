#!/usr/bin/env python3
"""A/B of the orbit-major tile order (option tile_order) on the workloads whose inputs are permuted
views of one buffer (GPU box only)."""
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
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3  # us


def main():
    rows = []
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    for n, reps in ((32, 300), (64, 100), (128, 5)):
        tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        rows.append((f"bcast4 {n}^4 f64", lambda a, b, c, d: a + b + c + d, (B,) + tuple(A.permutedims(q) for q in perms), 16 * n ** 4, reps))
        del tA, tB
    for m, reps in ((4000, 50), (8192, 20)):
        tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        rows.append((f"sym {m}^2 f64", lambda x, y: (x + y) / 2, (B, A, A.permutedims((1, 0))), 16 * m * m, reps))
    for name, f, arrays, algb, reps in rows:
        for order, tl in ((0, 0), (1, 0)):
            S.set_option("tile_order", order)
            S.set_option("tile_log2", tl)
            plan = S.make_plan(f, None, None, arrays[0].size, arrays)
            us = time_plan(plan, reps)
            d = plan.describe()
            print(f"{name:18s} tile_order={order} tl={tl:2d} {us:10.2f} us {algb / us / 1e3:8.1f} GB/s | {d[d.find('tile='):d.find(' algbytes')]}")
            sys.stdout.flush()
    S.set_option("tile_order", 1)
    S.set_option("tile_log2", 0)


if __name__ == "__main__":
    main()
