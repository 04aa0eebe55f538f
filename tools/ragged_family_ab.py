"""Ragged permutedims: the planner's choice against tiled_uavec = 0 (element-aligned vectors off) and flat = 0 (TILED forced).
Usage: python tools/ragged_family_ab.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import strided_jl_amd as S
from bench import colmajor_view, event_time_ms, graph_of
lib = S._lib.load()
cur = lambda: int(torch.cuda.current_stream().cuda_stream)
def mk(dims, dt=torch.float64):
    return colmajor_view(S, torch.randn(int(np.prod(dims)), dtype=dt, device="cuda"), dims)
def t(plan, reps=40):
    plan.execute(cur()); torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps); g.replay(); torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3
cases = [((7200, 100), (1, 0)), ((100, 7200), (1, 0)), ((7200, 104), (1, 0)), ((100, 100000), (1, 0)), ((100000, 100), (1, 0)), ((100, 90, 80), (1, 2, 0)), ((100, 90, 80), (2, 0, 1)), ((100, 90, 80, 7), (1, 0, 2, 3)), ((72, 72, 72, 5), (3, 2, 1, 0)), ((999, 1001), (1, 0)), ((1001, 999), (1, 0)), ((1000, 1000), (1, 0)), ((257, 129, 65), (2, 1, 0)), ((257, 129, 65), (1, 0, 2)), ((17, 33, 65, 31), (3, 2, 1, 0)), ((48, 36, 24, 30), (3, 2, 1, 0)), ((1000, 3, 700), (2, 1, 0)), ((100, 90, 80), (1, 0, 2)), ((100, 90, 80), (2, 1, 0))]
for dt in (torch.float64, torch.float32):
    for dims, p in cases:
        A = mk(dims, dt); B = mk(tuple(dims[i] for i in p), dt)
        res = []
        for opts in ({}, {"flat_wide": 0}, {"flat_wide": 2}):
            for k in ("flat", "flat2", "flatb", "tiled_uavec", "flat_wide"):
                try: S._lib.check(lib.smr_set_option(k.encode(), opts.get(k, 1)))
                except Exception: pass
            plan = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(p)))
            d = plan.describe()
            res.append("%6.2f us %-6s" % (t(plan), d[d.find("family=") + 7:d.find(" ct=")]))
        print("%-8s %-18s %-12s %5.1f MiB | default %s | flat_wide=0 %s | flat_wide=2 %s" % (str(dt)[6:], dims, p, plan.algorithmic_bytes / 2**20, *res))
