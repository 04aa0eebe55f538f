#!/usr/bin/env python3
"""Planner sanity sweep (GPU box only): many mid-sized workloads of mixed shape; prints effective GB/s and the chosen
family, slowest first -- anything far below its neighbours is a planner hole."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

fn = S.fn


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def t(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(3)) / reps * 1e3


def mk(dims, dt=torch.float64):
    n = int(np.prod(dims))
    return colmajor_view(S, torch.randn(n, dtype=dt, device="cuda"), dims)


rows = []


def rec(label, plan):
    n = plan.algorithmic_bytes
    reps = max(3, min(100, int(4e8 / max(n, 1))))
    us = t(plan, reps)
    d = plan.describe()
    rows.append((n / us / 1e3, f"{n / us / 1e3:8.1f} GB/s {us:9.2f} us {n / 2**20:8.1f} MiB | {label:44s} | {d[d.find('family='):d.find(' ct=')]} {d[d.find('dims='):][:70]}"))


# permutations of 3-/4-/5-d arrays with awkward extents
for dims in ((100, 90, 80), (257, 129, 65), (48, 36, 24, 30), (17, 33, 65, 31), (12, 10, 14, 9, 11), (1000, 3, 700), (3, 1000, 700), (640, 480, 3)):
    A = mk(dims)
    N = len(dims)
    perms = [p for p in itertools.permutations(range(N)) if p != tuple(range(N))]
    for p in (perms[:3] + perms[-2:]) if N > 3 else perms:
        B = mk(tuple(dims[i] for i in p))
        rec(f"permutedims {dims} {p}", S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(p))))
# ragged matrices: transposes and adjoints of (999, 1001)
for dt in (torch.float64, torch.float32):
    A = mk((999, 1001), dt)
    rec(f"transpose (999, 1001) {str(dt)[6:]}", S.make_plan(lambda x: x, None, None, (1001, 999), (mk((1001, 999), dt), A.permutedims((1, 0)))))
    Bs = mk((999, 999), dt)
    rec(f"B .= A[:, :999] .+ A[:, :999]' {str(dt)[6:]}", S.make_plan(lambda x, y: x + y, None, None, (999, 999), (Bs, A.sview(slice(None), slice(0, 999)), A.sview(slice(None), slice(0, 999)).permutedims((1, 0)))))
# tensor-network-like shapes: bond dims mixed with physical dims of 2..4
import random
random.seed(1)
for dims in ((64, 2, 64, 2, 16), (32, 4, 32, 4, 8), (16, 16, 4, 4, 16, 16), (2, 128, 2, 128, 8), (4, 4, 4, 4, 4, 4, 4, 4), (2, 2, 256, 2, 2, 256), (100, 3, 100, 3, 10)):
    A = mk(dims)
    N = len(dims)
    allp = list(itertools.permutations(range(N)))
    for p in [tuple(reversed(range(N)))] + random.sample(allp, 4):
        if p == tuple(range(N)):
            continue
        B = mk(tuple(dims[i] for i in p))
        rec(f"TN permutedims {dims} {p}", S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(p))))
# broadcasts with stride-0 operands
for dims in ((2000, 1500), (300, 200, 50)):
    A, B = mk(dims), mk(dims)
    col = mk((dims[0],) + (1,) * (len(dims) - 1))
    rowv = mk((1,) + dims[1:])
    arrs = S.promoteshape(dims, B, A, col, rowv)
    rec(f"B .= A .* col .+ row {dims}", S.make_plan(lambda a, c, r: a * c + r, None, None, dims, arrs))
# views: offsets, steps, reversed
A, B = mk((2048, 2048)), mk((2048, 2048))
rec("sub-box 1000x900 copy", S.make_plan(lambda x: x, None, None, (1000, 900), (B.sview(slice(5, 1005), slice(7, 907)), A.sview(slice(100, 1100), slice(50, 950)))))
rec("step-2 rows 1024x2048 *2", S.make_plan(lambda x: x * 2, None, None, (1024, 2048), (B.sview(slice(0, 1024), slice(None)), A.sview(slice(0, 2048, 2), slice(None)))))
rec("step-3 cols 2048x682 transpose-add", S.make_plan(lambda x, y: x + y, None, None, (682, 682), (B.sview(slice(0, 682), slice(0, 682)), A.sview(slice(0, 2046, 3), slice(0, 682)), A.permutedims((1, 0)).sview(slice(0, 682), slice(0, 2046, 3)))))
rec("reversed dim-0 copy 2048^2", S.make_plan(lambda x: x, None, None, (2048, 2048), (B, A.sview(slice(None, None, -1), slice(None)))))
rec("reversed dim-1 transpose 2048^2", S.make_plan(lambda x: x, None, None, (2048, 2048), (B, A.permutedims((1, 0)).sview(slice(None), slice(None, None, -1)))))
# partial reductions over each dim subset
for dims in ((512, 384, 64), (100, 90, 80, 7)):
    A = mk(dims, torch.float32)
    for k in range(1, len(dims)):
        for rd in itertools.combinations(range(len(dims)), k):
            out = A.similar(size=tuple(1 if d in rd else n for d, n in enumerate(dims)))
            rec(f"sum {dims} dims={rd}", S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A)))
# complex / f32 element-wise
for dt in (torch.float32, torch.complex64, torch.complex128):
    A, B, Cc = mk((1500, 1400), dt), mk((1500, 1400), dt), mk((1500, 1400), dt)
    rec(f"axpby {dt}", S.make_plan(lambda x, y: 2 * x + 3 * y, None, None, (1500, 1400), (Cc, A, B)))
    rec(f"adjoint {dt}", S.make_plan(lambda x: x, None, None, (1400, 1500), (mk((1400, 1500), dt), A.adjoint())))
for gbs, line in sorted(rows):
    print(line)
