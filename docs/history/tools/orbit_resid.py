#!/usr/bin/env python3
"""ORBIT 4-way sum at 32^4: does limiting the resident workgroups per CU (inflated LDS request) stagger loads and stores?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
for n in (32, 24, 40):
    tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
    row = []
    for ldsmin in (0, 20 << 10, 26 << 10, 32 << 10, 40 << 10, 53 << 10, 80 << 10):
        for pipe in (0, 1):
            S.set_option("orbit_lds_min", ldsmin)
            S.set_option("orbit_pipe", pipe)
            plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(p) for p in perms))
            plan.execute(cur())
            torch.cuda.synchronize()
            g = graph_of(torch, lambda: plan.execute(cur()), 300)
            g.replay()
            torch.cuda.synchronize()
            us = min(event_time_ms(torch, g.replay, 3) for _ in range(4)) / 300 * 1e3
            row.append(f"lds>={ldsmin >> 10}K pipe={pipe}: {us:6.2f}")
    print(f"sum4 {n}^4 f64 | " + " | ".join(row))
S.set_option("orbit_lds_min", 0)
S.set_option("orbit_pipe", -1)
