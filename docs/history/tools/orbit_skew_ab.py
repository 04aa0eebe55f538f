#!/usr/bin/env python3
"""ORBIT work list: natural enumeration of the super-cells against a diagonal (skewed) one, sizes around the power-of-two
collapse.  Usage: python tools/orbit_skew_ab.py [sizes...]"""
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
sizes = [int(a) for a in sys.argv[1:]] or [64, 96, 128, 144]
for dt in (torch.float64, torch.float32):
    for n in sizes:
        tA = torch.randn(n ** 4, dtype=dt, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        a4 = tA.reshape((n,) * 4)
        cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
        ref = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1) if n <= 96 else None
        row = []
        for grp, skew in ((2, 0), (2, 1), (2, 3), (2, 5), (1, 1), (1, 3), (4, 1)):
            S._lib.check(lib.smr_set_option(b"orbit_group", grp))
            S._lib.check(lib.smr_set_option(b"orbit_skew", skew))
            plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
            tB.zero_()
            us = time_plan(plan, 20 if n <= 64 else 3)
            ok = "" if ref is None or torch.equal(tB, ref) else " WRONG"
            row.append("g%d s%d %8.1f us %5.0f GB/s%s" % (grp, skew, us, 2 * tA.element_size() * n ** 4 / us / 1e3, ok))
        S._lib.check(lib.smr_set_option(b"orbit_group", 2))
        S._lib.check(lib.smr_set_option(b"orbit_skew", 0))
        print("sum4 %3d^4 %-8s | " % (n, str(dt)[6:]) + " | ".join(row))
        sys.stdout.flush()
        del tA, tB, a4, ref
        torch.cuda.empty_cache()
