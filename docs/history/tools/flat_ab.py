#!/usr/bin/env python3
"""Transposing copies with short leading dims whose extents are not powers of two: FLAT family (round 3) against the tiled
family (option flat=0).  Usage: python tools/flat_ab.py"""
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


cases = [((640, 480, 3), (2, 1, 0)), ((640, 480, 3), (2, 0, 1)), ((3, 480, 640), (2, 1, 0)), ((3, 1000, 700), (2, 1, 0)), ((3, 1000, 700), (1, 2, 0)),
         ((100, 3, 100, 3, 10), (4, 3, 2, 1, 0)), ((100, 3, 100, 3, 10), (1, 0, 4, 3, 2)), ((100, 3, 100, 3, 10), (4, 1, 0, 2, 3)),
         ((1920, 1080, 3), (2, 0, 1)), ((3, 1920, 1080), (1, 2, 0)), ((5, 300, 300, 7), (3, 2, 1, 0)), ((12, 10, 14, 9, 11), (4, 3, 2, 1, 0)),
         ((6, 2048, 2048), (2, 1, 0)), ((3, 1000, 700), (0, 2, 1)), ((3, 1920, 1080), (0, 2, 1)), ((6, 100, 50, 40), (0, 3, 2, 1)), ((5, 1024, 1024), (0, 2, 1))]
for dt in (torch.float64, torch.float32, torch.complex64):
    for shape, q in cases:
        N = 1
        for d in shape:
            N *= d
        tA = torch.randn(N, dtype=dt, device="cuda")
        tB = torch.empty_like(tA)
        A = colmajor_view(S, tA, shape)
        dshape = tuple(shape[i] for i in q)
        B = colmajor_view(S, tB, dshape)
        ref = tA.reshape(tuple(reversed(shape))).permute(*[len(shape) - 1 - q[len(shape) - 1 - i] for i in range(len(shape))]).contiguous().reshape(-1)
        row = []
        for flat in (1, 0):
            S._lib.check(lib.smr_set_option(b"flat", flat))
            plan = S.make_plan(lambda x: x, None, None, dshape, (B, A.permutedims(q)))
            tB.zero_()
            us = time_plan(plan, 50)
            ok = torch.equal(tB, ref)
            d = plan.describe()
            row.append("%-5s %7.2f us %5.0f GB/s%s" % (d[d.find("family=") + 7:d.find(" ct=")], us, 2 * tA.element_size() * N / us / 1e3, "" if ok else " WRONG"))
        S._lib.check(lib.smr_set_option(b"flat", 1))
        print("%-8s permutedims %-22s %-16s %5.1f MiB | %s" % (str(dt)[6:], shape, q, 2 * tA.element_size() * N / 2 ** 20, " | ".join(row)))
        sys.stdout.flush()
        del tA, tB
