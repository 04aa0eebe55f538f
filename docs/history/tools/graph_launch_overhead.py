#!/usr/bin/env python3
"""Fixed cost of ONE replay of a K-step graph inside a barrier-bracketed timed region (the driver times K = 20 steps):
torch's CUDAGraph.replay() + torch.cuda.synchronize() against an event spin before the synchronize, and the device-side
time of the same replay from HIP events.  Usage: python tools/graph_launch_overhead.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, graph_of  # noqa: E402

n = 32
tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
tB = torch.empty_like(tA)
tC = torch.empty_like(tA)
A, B, Cc = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
p3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def step():
    p2.execute(cur())
    p3.execute(cur())


for K in (20, 100, 500):
    g = graph_of(torch, step, K)
    g.replay()
    torch.cuda.synchronize()
    res = {}
    for mode in ("sync", "spin+sync", "stream-sync"):
        best = 1e9
        for _ in range(30):
            torch.cuda.synchronize()
            ev = torch.cuda.Event()
            t0 = time.perf_counter()
            g.replay()
            if mode == "spin+sync":
                ev.record()
                while not ev.query():
                    pass
                torch.cuda.synchronize()
            elif mode == "stream-sync":
                torch.cuda.current_stream().synchronize()
            else:
                torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        res[mode] = best * 1e6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dev = 1e9
    for _ in range(10):
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        dev = min(dev, e0.elapsed_time(e1) * 1e3)
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    host = (time.perf_counter() - t0) / 20 * 1e6
    torch.cuda.synchronize()
    print("K=%3d: wall per replay (us): %s | device (HIP events) %.1f | host cost of replay() alone %.1f | per step: %s" % (
        K, ", ".join("%s %.1f" % kv for kv in res.items()), dev, host, ", ".join("%s %.2f" % (k, v / K) for k, v in res.items())))
