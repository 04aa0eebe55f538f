"""Transposes of ragged / line-misaligned shapes against their aligned neighbours: us per launch in a hipGraph.
Usage: python tools/ragged_cases.py [option=value ...]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import strided_jl_amd as S
from bench import colmajor_view, event_time_ms, graph_of
lib = S._lib.load()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    S._lib.check(lib.smr_set_option(k.encode(), int(v)))
cur = lambda: int(torch.cuda.current_stream().cuda_stream)
def mk(dims, dt=torch.float64):
    return colmajor_view(S, torch.randn(int(np.prod(dims)), dtype=dt, device="cuda"), dims)
def t(plan, reps=40):
    plan.execute(cur()); torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps); g.replay(); torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3
for dims, p in (((7200, 100), (1, 0)), ((7168, 128), (1, 0)), ((7200, 96), (1, 0)), ((7200, 128), (1, 0)), ((7168, 100), (1, 0)), ((7200, 104), (1, 0)), ((100, 7200), (1, 0)), ((128, 7168), (1, 0)),
                ((80, 9000), (1, 0)), ((96, 9216), (1, 0)), ((80, 9216), (1, 0)), ((96, 9000), (1, 0)),
                ((100, 90, 80), (1, 0, 2)), ((96, 96, 80), (1, 0, 2)), ((100, 90, 80), (2, 1, 0)), ((96, 96, 96), (2, 1, 0)), ((999, 1001), (1, 0)), ((1024, 1024), (1, 0)), ((992, 1024), (1, 0)), ((1000, 1000), (1, 0))):
    A = mk(dims); B = mk(tuple(dims[i] for i in p))
    plan = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(p)))
    us = t(plan); d = plan.describe(); n = plan.algorithmic_bytes
    print("%-16s %-10s %7.2f us %6.0f GB/s %5.1f MiB | %s" % (dims, p, us, n / us / 1e3, n / 2**20, d[d.find("family="):d.find(" ct=")] + " " + d[d.find("dims="):d.find(" algbytes")]))
