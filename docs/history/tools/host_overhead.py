#!/usr/bin/env python3
"""Host-side cost per call of the Python front end (GPU box only): the one-shot funnel
`_mapreduce_fuse_` (trace cache hit + smr_mapreduce plan-cache hit), a prepared plan's execute(), and the
same inside a hipGraph replay (no host work per kernel).  Kernel: permutedims! of a 64x64 Float64 matrix
(negligible GPU time), so the numbers are host time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, graph_of  # noqa: E402


def main():
    n = 64
    tA = torch.randn(n * n, dtype=torch.float64, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (n, n)), colmajor_view(S, tB, (n, n))
    src = A.permutedims((1, 0))
    f = lambda x: x  # noqa: E731
    s = int(torch.cuda.current_stream().cuda_stream)
    iters = 20000

    def bench(name, fn):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name:46s} host {1e6 * (t1 - t0) / iters:7.2f} us/call (enqueue), {1e6 * (t2 - t0) / iters:7.2f} us/call incl. drain")

    bench("_mapreduce_fuse_ (trace cache + plan cache hit)", lambda: S._mapreduce_fuse_(f, None, None, B.size, (B, src)))
    bench("same, a fresh lambda every call (re-trace)", lambda: S._mapreduce_fuse_(lambda x: x, None, None, B.size, (B, src)))
    plan = S.make_plan(f, None, None, B.size, (B, src))
    bench("Plan.execute", lambda: plan.execute(s))
    g = graph_of(torch, lambda: plan.execute(int(torch.cuda.current_stream().cuda_stream)), 500)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"{'hipGraph replay of 500 launches':46s} {1e6 * (t1 - t0) / (20 * 500):7.2f} us/launch end to end")


if __name__ == "__main__":
    main()
