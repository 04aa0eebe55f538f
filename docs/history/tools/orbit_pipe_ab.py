#!/usr/bin/env python3
"""ORBIT kernel: one-shot workgroups vs the persistent pipelined form (option orbit_pipe 0 / 1 / -1), GPU box only."""
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
    return min(event_time_ms(torch, g.replay, 3) for _ in range(4)) / reps * 1e3


def main():
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    cases = []
    for dt in (torch.float64, torch.float32):
        for n in (32, 48, 64, 80, 96, 128):
            if dt == torch.float32 and n < 64:
                continue
            tA = torch.randn(n ** 4, dtype=dt, device="cuda")
            tB = torch.empty_like(tA)
            A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
            reps = 300 if n <= 32 else (40 if n <= 64 else 4)
            cases.append((f"sum4 {n}^4 {str(dt)[6:]}", lambda a, b, c, d: a + b + c + d, A.size, (B,) + tuple(A.permutedims(p) for p in perms), reps, (tA, tB)))
    for m in (4000, 8192, 16384):
        tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        cases.append((f"sym {m}^2 f64", lambda x, y: (x + y) / 2, (m, m), (B, A, A.adjoint()), 10, (tA, tB)))
    for label, f, dims, arrs, reps, keep in cases:
        row, outs = [], []
        for pipe in (0, 1, -1):
            S.set_option("orbit_pipe", pipe)
            plan = S.make_plan(f, None, None, dims, arrs)
            keep[1].zero_()
            us = time_plan(plan, reps)
            outs.append(keep[1].clone())
            row.append(f"pipe={pipe}: {us:9.2f} us {plan.algorithmic_bytes / us / 1e3:7.1f} GB/s")
        d = plan.describe()
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print(f"{label:22s} | " + " | ".join(row) + f" | {'same' if same else 'DIFFERENT'} | {d[d.find('tile='):d.find(' algb')]}")
        sys.stdout.flush()
    S.set_option("orbit_pipe", -1)


if __name__ == "__main__":
    main()
