#!/usr/bin/env python3
"""ORBIT family (4-way permuted sum) over sizes around the power-of-two collapse, with the work list grouped by
super-cells of 1 (natural), 2 (round-2 default) and 4 tiles per dim.  Usage: python tools/orbit_group_ab.py [sizes...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

lib = S._lib.load()


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(4)) / reps * 1e3


perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
sizes = [int(a) for a in sys.argv[1:]] or [64, 96, 112, 120, 128, 136, 144]
for dt in (torch.float64, torch.float32):
    for n in sizes:
        if dt == torch.float32 and n not in (64, 96, 128):
            continue
        tA = torch.randn(n ** 4, dtype=dt, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        reps = 30 if n <= 64 else (6 if n <= 100 else 3)
        row = []
        for grp in (1, 2, 4, 8):
            S._lib.check(lib.smr_set_option(b"orbit_group", grp))
            plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
            us = time_plan(plan, reps)
            row.append(f"group {grp}: {us:9.2f} us {2 * tA.element_size() * n ** 4 / us / 1e3:7.1f} GB/s")
            d = plan.describe()
        S._lib.check(lib.smr_set_option(b"orbit_group", 2))
        print(f"sum4 {n:4d}^4 {str(dt)[6:]:8s} | " + " | ".join(row) + " | " + d[d.find("tile="):d.find(" algb")])
        sys.stdout.flush()
        del tA, tB, A, B, plan
        torch.cuda.empty_cache()
