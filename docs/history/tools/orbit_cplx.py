#!/usr/bin/env python3
"""ComplexF32 / ComplexF64 4-way permuted sum: persistent pipelined ORBIT form against the one-shot form (the c32 pipelined
kernel spills 20 bytes of scratch at 128 VGPRs).  Usage: python tools/orbit_cplx.py"""
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
    return min(event_time_ms(torch, g.replay, 3) for _ in range(5)) / reps * 1e3


perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
for dt in (torch.complex64, torch.float64, torch.complex128):
    for n in (64, 80, 96, 128):
        if dt == torch.complex128 and n > 96:
            continue
        tA = torch.randn(n ** 4, dtype=dt, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        row = []
        for pipe in (-1, 0, 1):
            S._lib.check(lib.smr_set_option(b"orbit_pipe", pipe))
            plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
            us = time_plan(plan, 20 if n <= 64 else 4)
            row.append("pipe=%2d %9.2f us %5.0f GB/s" % (pipe, us, 2 * tA.element_size() * n ** 4 / us / 1e3))
        S._lib.check(lib.smr_set_option(b"orbit_pipe", -1))
        d = plan.describe()
        print("sum4 %3d^4 %-10s | " % (n, str(dt)[6:]) + " | ".join(row) + " | " + d[d.find("tile="):d.find(" algb")])
        sys.stdout.flush()
        del tA, tB
        torch.cuda.empty_cache()
