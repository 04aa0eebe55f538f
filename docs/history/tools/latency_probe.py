#!/usr/bin/env python3
"""Is the 4-way sum at 32^4 bound by the latency of one workgroup's phase chain or by throughput?
Same per-workgroup work on fewer / more workgroups (GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps=300):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3


def main():
    for n in (16, 32, 48, 64):
        tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
        for tl in (10, 12):
            S.set_option("tile_log2", tl)
            plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
            us = time_plan(plan, 100 if n > 32 else 300)
            d = plan.describe()
            print(f"bcast4 {n}^4 tl={tl} {us:8.2f} us {16 * n ** 4 / us / 1e3:8.1f} GB/s | {d[d.find('tile='):d.find(' algbytes')]}")
        S.set_option("tile_log2", 0)
        plan = S.make_plan(lambda a: a, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
        us = time_plan(plan, 100 if n > 32 else 300)
        print(f"perm   {n}^4       {us:8.2f} us {16 * n ** 4 / us / 1e3:8.1f} GB/s")
        plan = S.make_plan(lambda a: a, None, None, A.size, (B, A))
        us = time_plan(plan, 100 if n > 32 else 300)
        print(f"copy   {n}^4       {us:8.2f} us {16 * n ** 4 / us / 1e3:8.1f} GB/s")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
