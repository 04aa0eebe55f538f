#!/usr/bin/env python3
"""HBM-cold 4-way permuted sum (rotating through more array pairs than the Infinity Cache holds, as bench.py's `cold` leg does):
the ORBIT family (A read once, 32- / 64-byte rows) against the classic TILED kernel (A read through four tiles, longer rows) --
which one should a caller that KNOWS its data is cold ask for (option "orbit")?  Warm numbers beside them.
Usage: python tools/cold_orbit_vs_tiled.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


print(torch.cuda.get_device_name(0), "-- C .= sum of 4 permuted views of A, Float64; us per launch (GB/s effective)")
for n, npair in ((32, 40), (48, 16), (64, 6)):
    poolA = torch.randn(npair, n ** 4, dtype=torch.float64, device="cuda")
    poolB = torch.empty_like(poolA)
    A, B = colmajor_view(S, poolA[0], (n,) * 4), colmajor_view(S, poolB[0], (n,) * 4)
    esz = 8 * n ** 4
    row = []
    for orbit in (1, 0):
        S.set_option("orbit", orbit)
        p = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
        p.execute(cur())
        torch.cuda.synchronize()
        state = {"i": 0}

        def rot():
            i = state["i"] % npair
            state["i"] += 1
            p.execute(cur(), bases=[poolB.data_ptr() + i * esz] + [poolA.data_ptr() + i * esz] * 4)

        reps = max(npair * 2, 24)
        g = graph_of(torch, rot, reps)
        g.replay()
        torch.cuda.synchronize()
        cold = min(event_time_ms(torch, g.replay, 2) for _ in range(3)) / reps * 1e3
        g2 = graph_of(torch, lambda: p.execute(cur()), 50)
        g2.replay()
        torch.cuda.synchronize()
        warm = min(event_time_ms(torch, g2.replay, 2) for _ in range(3)) / 50 * 1e3
        d = p.describe()
        row.append((orbit, cold, warm, d[d.find("family="):d.find(" ct=")], poolB[0].clone()))
    S.set_option("orbit", 1)
    same = bool(torch.equal(row[0][4], row[1][4]))
    by = 2 * esz
    footprint = 2 * npair * esz / 2 ** 20
    print("n = %d (%.0f MiB per array, %d pairs = %.0f MiB rotating): " % (n, esz / 2 ** 20, npair, footprint) +
          " | ".join("%s cold %.2f us (%.0f) warm %.2f us (%.0f)" % (r[3], r[1], by / r[1] / 1e3, r[2], by / r[2] / 1e3) for r in row) +
          " | results identical: %s" % same)
