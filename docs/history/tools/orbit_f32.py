#!/usr/bin/env python3
"""Float32 4-way permuted sum: 8^4 cubes (32-byte runs, few orbits at small sizes) against 4^4 cubes (16-byte runs, option
orbit_minrun=16) and the classic tiled kernel.  Usage: python tools/orbit_f32.py"""
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
for dt in (torch.float32, torch.complex64):
    for n in (16, 24, 32, 40, 48, 64, 96, 128):
        if dt == torch.complex64 and n > 64:
            continue
        tA = torch.randn(n ** 4, dtype=dt, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        a4 = tA.reshape((n,) * 4)
        cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
        ref = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)
        row = []
        for name, kw in (("default", {}), ("minrun 16", dict(orbit_minrun=16)), ("minrun 16, lg 2", dict(orbit_minrun=16, orbit_lg=2)),
                         ("lg 3", dict(orbit_lg=3)), ("tiled", dict(orbit=0))):
            setopt(orbit_minrun=32, orbit_lg=-1, orbit=1)
            setopt(**kw)
            try:
                plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
                tB.zero_()
                us = time_plan(plan, 200 if n <= 32 else (30 if n <= 64 else 4))
                ok = torch.equal(tB, ref)
                d = plan.describe()
                row.append("%s %8.2f us %5.0f GB/s%s [%s]" % (name, us, 2 * tA.element_size() * n ** 4 / us / 1e3, "" if ok else " WRONG",
                                                             d[d.find("family=") + 7:d.find(" ct=")] + " " + d[d.find("tile="):d.find(" group")].replace("tile=", "")[:28]))
            except Exception as e:  # noqa: BLE001
                row.append("%s failed (%s)" % (name, str(e)[:40]))
        setopt(orbit_minrun=32, orbit_lg=-1, orbit=1)
        print("sum4 %3d^4 %-9s | " % (n, str(dt)[6:]) + " | ".join(row))
        sys.stdout.flush()
        del tA, tB
        torch.cuda.empty_cache()
