#!/usr/bin/env python3
"""Runs the 4-way permuted sum (ORBIT family) eagerly at several sizes in ONE process, for rocprofv3 PMC passes that
compare a power-of-two size with its neighbours (tools/pmc_orbit_sizes.sh; the summaries are split by grid size).
Usage: python tools/prof_orbit_sizes.py [--sizes 96,128] [--iters 3] [--dtype f64] [--perm 0]
--perm 1 profiles permutedims!(B, A, (4,3,2,1)) (TILED family) instead."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="96,128")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--perm", type=int, default=0)
ap.add_argument("--opt", action="append", default=[], help="name=value library options")
args = ap.parse_args()
lib = S._lib.load()
for o in args.opt:
    k, v = o.split("=")
    S._lib.check(lib.smr_set_option(k.encode(), int(v)))
dt = torch.float64 if args.dtype == "f64" else torch.float32
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
s = int(torch.cuda.current_stream().cuda_stream)
for n in [int(x) for x in args.sizes.split(",")]:
    tA = torch.randn(n ** 4, dtype=dt, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
    if args.perm:
        p = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    else:
        p = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
    print(n, p.describe())
    for _ in range(args.iters):
        p.execute(s)
    torch.cuda.synchronize()
    del tA, tB, A, B, p
    torch.cuda.empty_cache()
