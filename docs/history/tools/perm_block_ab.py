#!/usr/bin/env python3
"""True-HBM transposes / permutedims! with the tiles in natural order against compact blocks (tile_block, experiment
tile_block_min_axes=2).  Usage: python tools/perm_block_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

lib = S._lib.load()


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(4)) / reps * 1e3


def setopt(**kw):
    for k, v in kw.items():
        S._lib.check(lib.smr_set_option(k.encode(), int(v)))


variants = [("natural", dict(tile_block=0)), ("blocks 2", dict(tile_block=2, tile_block_min_axes=2, tile_block_xcd=0)),
            ("blocks 4", dict(tile_block=4, tile_block_min_axes=2, tile_block_xcd=0)), ("blocks 8", dict(tile_block=8, tile_block_min_axes=2, tile_block_xcd=0)),
            ("blocks 4, XCD runs", dict(tile_block=4, tile_block_min_axes=2, tile_block_xcd=1)),
            ("natural, never persistent", dict(tile_block=0, tiled_persist=0))]
DT = {"f64": torch.float64, "f32": torch.float32, "c64": torch.complex64, "c128": torch.complex128}[os.environ.get("DT", "f64")]
cases = []
for n in (96, 128, 144):
    cases.append(("permutedims!(4,3,2,1) %d^4" % n, (n,) * 4, (3, 2, 1, 0)))
cases += [("permutedims!(2,3,4,1) 128^4", (128,) * 4, (1, 2, 3, 0)), ("permutedims!(3,4,1,2) 128^4", (128,) * 4, (2, 3, 0, 1)),
          ("transpose 8192^2", (8192, 8192), (1, 0)), ("transpose 16384^2", (16384, 16384), (1, 0)), ("transpose 12000^2", (12000, 12000), (1, 0))]
for name, shape, q in cases:
    N = 1
    for d in shape:
        N *= d
    tA = torch.randn(N, dtype=DT, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, shape), colmajor_view(S, tB, shape)
    row = []
    for vn, kw in variants:
        setopt(tile_block=-1, tile_block_min_axes=3, tile_block_xcd=-1, tiled_persist=1)
        setopt(**kw)
        plan = S.make_plan(lambda x: x, None, None, tuple(shape[i] for i in q), (colmajor_view(S, tB, tuple(shape[i] for i in q)), A.permutedims(q)))
        us = time_plan(plan, 5)
        row.append("%s %8.1f us %6.0f GB/s" % (vn, us, 2 * tA.element_size() * N / us / 1e3))
    setopt(tile_block=-1, tile_block_min_axes=3, tile_block_xcd=-1, tiled_persist=1)
    d = plan.describe()
    print("%-30s %-5s | " % (name, os.environ.get("DT", "f64")) + " | ".join(row) + " | " + d[d.find("tile="):d.find(" algb")])
    sys.stdout.flush()
    del tA, tB, A, B
    torch.cuda.empty_cache()
