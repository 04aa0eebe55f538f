#!/usr/bin/env python3
"""Runs the two headline launches eagerly N times (for rocprofv3 kernel traces / PMC passes).
Usage: python tools/prof_headline.py [--n 32] [--iters 200] [--which both|perm|bcast|copy]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--which", default="both")
args = ap.parse_args()
n = args.n
tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
tB = torch.empty_like(tA)
tC = torch.empty_like(tA)
A, B, C = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
plans = {
    "perm": S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0)))),
    "bcast": S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in perms)),
    "copy": S.make_plan(lambda x: x, None, None, A.size, (B, A)),
}
sel = ["perm", "bcast"] if args.which == "both" else [args.which]
s = int(torch.cuda.current_stream().cuda_stream)
for name in sel:
    print(name, plans[name].describe())
for _ in range(args.iters):
    for name in sel:
        plans[name].execute(s)
torch.cuda.synchronize()
