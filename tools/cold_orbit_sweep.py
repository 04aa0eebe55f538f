#!/usr/bin/env python3
"""Round 6: the HBM-cold 4-way permuted sum under the ORBIT family's work-list options -- does another tile
edge / super-cell edge / deal win COLD even where it loses warm?  Rotating through more (A, C) pairs than the 256-MiB Infinity
Cache holds (as bench.py's `cold` leg), hipGraph, HIP events; warm numbers beside them.
Usage: python tools/cold_orbit_sweep.py [n ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
DEFAULTS = {"orbit": 1, "orbit_lg": -1, "orbit_group": 2, "orbit_deal": 0, "orbit_pack": 1, "nt_store": -1, "orbit_pair": 1}


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


sizes = [int(a) for a in sys.argv[1:]] or [32, 48, 64]
print(torch.cuda.get_device_name(0), "-- C .= sum of 4 permuted views of A, Float64; us per launch (frac of 8 TB/s)")
for n in sizes:
    npair = max(3, int(700 * 2 ** 20 / (2 * 8 * n ** 4)) + 1)
    poolA = torch.randn(npair, n ** 4, dtype=torch.float64, device="cuda")
    poolB = torch.empty_like(poolA)
    A, B = colmajor_view(S, poolA[0], (n,) * 4), colmajor_view(S, poolB[0], (n,) * 4)
    esz = 8 * n ** 4
    by = 2 * esz
    variants = [("default", {}), ("group 1", {"orbit_group": 1}), ("group 4", {"orbit_group": 4}), ("deal 1", {"orbit_deal": 1}),
                ("group 4 deal 1", {"orbit_group": 4, "orbit_deal": 1}), ("nt stores", {"nt_store": 1}), ("pack 0", {"orbit_pack": 0}),
                ("pair form", {"orbit_pair": 2}), ("pair form, nt stores", {"orbit_pair": 2, "nt_store": 1}), ("classic tiled", {"orbit": 0})]
    if n % 8 == 0 and n <= 40:
        variants.insert(1, ("8^4 cubes", {"orbit_lg": 3}))
    if n % 4 == 0 and n > 40:
        variants.insert(1, ("4^4 cubes", {"orbit_lg": 2}))
    ref = None
    print("n = %d (%d pairs = %.0f MiB rotating)" % (n, npair, 2 * npair * esz / 2 ** 20))
    for name, opts in variants:
        for k, v in DEFAULTS.items():
            S.set_option(k, opts.get(k, v))
        p = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
        p.execute(cur())
        torch.cuda.synchronize()
        out = poolB[0].clone()
        if ref is None:
            ref = out
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
        print("   %-16s cold %8.2f us (%.3f) | warm %8.2f us (%.3f) | %s | %s" %
              (name, cold, by / cold / 1e3 / 8000, warm, by / warm / 1e3 / 8000, "same result" if torch.equal(out, ref) else "RESULT DIFFERS",
               d[d.find("family="):d.find(" algbytes")].replace(" ct=f64 f=add4 N=4 M=5", "")), flush=True)
        del p, g, g2
    for k, v in DEFAULTS.items():
        S.set_option(k, v)
    del poolA, poolB
