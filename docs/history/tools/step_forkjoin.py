#!/usr/bin/env python3
"""The bench step (permutedims! into B, 4-way permuted sum into C; both only READ A) captured three ways:
  in order      one stream, the second kernel waits for the first (what bench.py did in rounds 1-2)
  fork/join     the two kernels of a step are independent branches of the graph, joined at the end of every step
  two chains    one stream per output array (the launches of one output stay ordered), one join at the end
Timed with HIP events over graph replays; results checked against torch.  Usage: python tools/step_forkjoin.py [--n 32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()
n, R = args.n, args.reps
dev = torch.device("cuda", 0)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
tB = torch.empty_like(tA)
tC = torch.empty_like(tA)
A, B, Cc = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
p3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))
p2.execute(int(torch.cuda.current_stream().cuda_stream))
p3.execute(int(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
a4 = tA.reshape((n,) * 4)
ref2 = a4.permute(3, 2, 1, 0).contiguous().reshape(-1)
cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
ref3 = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)


def capture(variant):
    main = torch.cuda.Stream()
    side = torch.cuda.Stream()
    main.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        sm, ss = int(main.cuda_stream), int(side.cuda_stream)
        if variant == "in order":
            for _ in range(R):
                p2.execute(sm)
                p3.execute(sm)
        elif variant == "fork/join every step":
            for _ in range(R):
                side.wait_stream(main)
                p2.execute(sm)
                p3.execute(ss)
                main.wait_stream(side)
        else:
            side.wait_stream(main)
            for _ in range(R):
                p2.execute(sm)
                p3.execute(ss)
            main.wait_stream(side)
    return g, main


for variant in ("in order", "fork/join every step", "two chains, one join"):
    tB.zero_()
    tC.zero_()
    g, main = capture(variant)
    g.replay()
    torch.cuda.synchronize()
    ok = torch.equal(tB, ref2) and torch.equal(tC, ref3)
    us = min(event_time_ms(torch, g.replay, 2) for _ in range(7)) / R * 1e3
    gbs = 2 * 2 * 8 * n ** 4 / us * 1e-3
    print("step %-22s: %7.3f us per step  %7.1f GB/s (%.1f %% of 8 TB/s) %s" % (variant, us, gbs, gbs / 80.0, "ok" if ok else "WRONG"))
