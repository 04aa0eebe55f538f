#!/usr/bin/env python3
"""Runs the kernels of the BASELINE configs eagerly N times (for rocprofv3 kernel traces / PMC passes).
Usage: python tools/prof_headline.py [--n 32] [--iters 200] [--which both|perm|bcast|copy|c1|c4|c5]"""
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
ap.add_argument("--opt", action="append", default=[], help="name=value library option, may repeat")
args = ap.parse_args()
for kv in args.opt:
    k, v = kv.split("=")
    S.set_option(k, int(v))
n = args.n
fn = S.fn
plans = {}
keep = []
if args.which in ("both", "perm", "bcast", "copy"):
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
elif args.which == "c1":   # configs[0] on the GPU: B .= (A .+ A')./2, 4000x4000 Float64
    m = 4000
    tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
    plans["c1"] = S.make_plan(lambda x, y: (x + y) / 2, None, None, (m, m), (B, A, A.adjoint()))
elif args.which == "c4":   # configs[3], one GPU: mapreduce(abs2, +, A), 4096x4096x64 Float32
    tA = torch.rand(4096 * 4096 * 64, dtype=torch.float32, device="cuda") * 2 - 1
    A = colmajor_view(S, tA, (4096, 4096, 64))
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    O = S.StridedView(out, A.size, (0, 0, 0), 0)
    plans["c4"] = S.make_plan(fn.abs2, "+", None, A.size, (O, A))
    keep.append(out)
elif args.which == "c5":   # configs[4]: B .= A.*exp.(-2A) .+ sin.(A.*A), 8192x8192 Float32
    m = 8192
    tA = torch.rand(m * m, dtype=torch.float32, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
    plans["c5"] = S.make_plan(lambda a: a * fn.exp(-2 * a) + fn.sin(a * a), None, None, (m, m), (B, A))
sel = ["perm", "bcast"] if args.which == "both" else [args.which]
s = int(torch.cuda.current_stream().cuda_stream)
for name in sel:
    print(name, plans[name].describe())
for _ in range(args.iters):
    for name in sel:
        plans[name].execute(s)
torch.cuda.synchronize()
