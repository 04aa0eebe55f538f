#!/usr/bin/env python3
"""A/B of the XCD-class tile dealing (option xcd_classes) on workloads whose inputs are
dim-permuted views of one array."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402
from tune import time_plan  # noqa: E402


def run(name, f, arrays, reps, algb, opts=()):
    for cls in (0, 1):
        for k, v in opts:
            S.set_option(k, v)
        S.set_option("xcd_classes", cls)
        plan = S.make_plan(f, None, None, arrays[0].size, arrays)
        us = time_plan(plan, reps)
        print(f"{name:28s} xcd_classes={cls} {us:10.2f} us {algb / us / 1e3:8.1f} GB/s | {plan.describe()[:110]}")
    for k, v in opts:
        S.set_option(k, -1 if k.startswith("tile_lg") else 0)
    S.set_option("max_lds_bytes", 65536)


perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
add4 = lambda a, b, c, d: a + b + c + d  # noqa: E731
for n, reps in ((32, 200), (64, 50), (128, 5)):
    tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
    arrs = (B,) + tuple(A.permutedims(q) for q in perms)
    run(f"bcast4 {n}^4 tile 8,8,4,4", add4, arrs, reps, 16 * n ** 4)
    run(f"bcast4 {n}^4 tile 8^4/1024thr", add4, arrs, reps, 16 * n ** 4,
        (("max_lds_bytes", 160 * 1024), ("tile_log2", 12), ("tile_lg0", 3), ("tile_lg1", 3), ("tile_lg2", 3), ("tile_lg3", 3)))
    del tA, tB
for m, reps in ((1024, 200), (4000, 50), (16384, 5)):
    tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
    run(f"symmetrise {m}^2", lambda x, y: (x + y) / 2, (B, A, A.adjoint()), reps, 16 * m * m)
    del tA, tB
