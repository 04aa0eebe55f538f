#!/usr/bin/env python3
"""C .= beta .* C .+ alpha .* permutedims(A, p) (TensorOperations' tensoradd!) and B .= A1 .+ permutedims(A2, p) on shapes whose
unit-stride dims are short and not powers of two: FLAT family (one- and two-sided forms, n-ary since round 3) against the TILED family
(option flat = 0).  Usage: python tools/flat_nary_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

lib = S._lib.load()


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(5)) / reps * 1e3


cases = [((5, 300, 300, 7), (3, 2, 1, 0)), ((5, 300, 300, 7), (3, 1, 2, 0)), ((640, 480, 3), (2, 1, 0)), ((3, 480, 640), (2, 1, 0)), ((3, 1000, 700), (1, 2, 0)),
         ((17, 33, 65, 31), (3, 2, 1, 0)), ((6, 64, 64, 64, 5), (4, 3, 2, 1, 0)), ((100, 3, 100, 3, 10), (4, 3, 2, 1, 0)), ((10, 200, 200, 10), (3, 2, 1, 0)),
         ((1920, 1080, 3), (2, 0, 1))]
for dt in (torch.float64, torch.float32, torch.complex128):
    for shape, q in cases:
        N = 1
        for d in shape:
            N *= d
        n = len(shape)
        tA = torch.randn(N, dtype=dt, device="cuda")
        tC0 = torch.randn(N, dtype=dt, device="cuda")
        tC = tC0.clone()
        A = colmajor_view(S, tA, shape)
        dshape = tuple(shape[i] for i in q)
        Cv = colmajor_view(S, tC, dshape)
        pa = tA.reshape(tuple(reversed(shape))).permute(*[n - 1 - q[n - 1 - i] for i in range(n)]).contiguous().reshape(-1)
        ref = 0.5 * tC0 + 2.0 * pa
        row = []
        for flat in (0, 1):
            S._lib.check(lib.smr_set_option(b"flat", flat))
            # constants of the arrays' own precision (a Float64 constant would promote Float32 arrays to a Float64 compute type, as in Julia)
            h, w = (np.float32(0.5), np.float32(2.0)) if dt == torch.float32 else (0.5, 2.0)
            plan = S.make_plan(lambda c, a: h * c + w * a, None, None, dshape, (Cv, Cv, A.permutedims(q)))
            tC.copy_(tC0)
            plan.execute(cur())
            torch.cuda.synchronize()
            ok = torch.allclose(tC, ref, rtol=1e-6 if dt == torch.float32 else 1e-13, atol=0)
            us = time_plan(plan, 30)
            d = plan.describe()
            lab = "flat2" if "two-sided" in d else d[d.find("family=") + 7:d.find(" ct=")]
            row.append("%-5s %7.2f us %5.0f GB/s%s" % (lab, us, 3 * tA.element_size() * N / us / 1e3, "" if ok else " WRONG"))
        S._lib.check(lib.smr_set_option(b"flat", 1))
        print("%-10s C .= C/2 .+ 2 .* permutedims(A) %-20s %-16s %6.1f MiB | %s" % (str(dt)[6:], shape, q, 3 * tA.element_size() * N / 2 ** 20, " | ".join(row)))
        sys.stdout.flush()
        del tA, tC, tC0
