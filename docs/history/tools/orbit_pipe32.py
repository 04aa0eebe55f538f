#!/usr/bin/env python3
"""4-way permuted sum at cache-resident sizes: one orbit per workgroup (round 2) against the persistent pipelined form
with 2-4 orbits per workgroup (orbit_pipe=1 + a grid cap through orbit_lds_min / orbit_wgs).  Usage: python tools/orbit_pipe32.py"""
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
    return min(event_time_ms(torch, g.replay, 3) for _ in range(6)) / reps * 1e3


def setopt(**kw):
    for k, v in kw.items():
        S._lib.check(lib.smr_set_option(k.encode(), int(v)))


perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
for dt in (torch.float64, torch.float32):
    for n in (16, 24, 32, 40, 48):
        tA = torch.randn(n ** 4, dtype=dt, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        a4 = tA.reshape((n,) * 4)
        cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
        ref = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)
        row = []
        for name, kw in (("one-shot", dict(orbit_pipe=0, orbit_wgs=0)), ("pipe 1024", dict(orbit_pipe=1, orbit_wgs=1024)), ("pipe 768", dict(orbit_pipe=1, orbit_wgs=768)),
                         ("pipe 512", dict(orbit_pipe=1, orbit_wgs=512)), ("pipe 384", dict(orbit_pipe=1, orbit_wgs=384)), ("pipe 256", dict(orbit_pipe=1, orbit_wgs=256))):
            setopt(**kw)
            plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
            tB.zero_()
            us = time_plan(plan, 200 if n <= 32 else 50)
            ok = torch.equal(tB, ref)
            row.append("%s %6.2f%s" % (name, us, "" if ok else " WRONG"))
            d = plan.describe()
        setopt(orbit_pipe=-1, orbit_wgs=0)
        print("sum4 %3d^4 %-8s | " % (n, str(dt)[6:]) + " | ".join(row) + " | " + d[d.find("family="):d.find(" ct=")] + " " + d[d.find("tile="):d.find(" algb")])
        sys.stdout.flush()
