#!/usr/bin/env python3
"""Planner cliff of the general multi-stride map: inputs with DIFFERENT unit axes that are
distinct arrays (ORBIT does not apply) through the TILED family with small tiles (1024 elements / 256 lanes), big
tiles (4096 / 1024 lanes, option tile_log2=12) and the persistent pipelined form, at sizes around the old
256 < n/4096 < 8192 window.  Also the 3-array case B .= A .+ permutedims(C, ...) .* D'-style.
Usage: python tools/cliff_ab.py [sizes...]"""
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


perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
sizes = [int(a) for a in sys.argv[1:]] or [32, 48, 64, 96, 128]
variants = [("default (auto)", {}), ("round 2 (natural order)", dict(tile_block=0)),
            ("small, natural", dict(tile_log2=10, tile_block=0)), ("big, natural", dict(tile_log2=12, tile_block=0)),
            ("big, blocks of 3", dict(tile_log2=12, tile_block=3)), ("big, blocks of 4", dict(tile_log2=12, tile_block=4)),
            ("big, blocks of 6", dict(tile_log2=12, tile_block=6)), ("big, run-balanced blocks", dict(tile_log2=12, tile_block=-2)),
            ("big, run-balanced, XCD runs", dict(tile_log2=12, tile_block=-2, tile_block_xcd=1)), ("small, run-balanced blocks", dict(tile_log2=10, tile_block=-2)),
            ("big, blocks of 4, XCD runs", dict(tile_log2=12, tile_block=4, tile_block_xcd=1)),
            ("big, blocks of 2, XCD runs", dict(tile_log2=12, tile_block=2, tile_block_xcd=1)),
            ("small, blocks of 4", dict(tile_log2=10, tile_block=4)), ("small, blocks of 6", dict(tile_log2=10, tile_block=6)),
            ("small, blocks of 4, XCD runs", dict(tile_log2=10, tile_block=4, tile_block_xcd=1))]
defaults = dict(tile_log2=0, tiled_persist=1, tiled_persist_min=32, tile_block=-1, tile_block_xcd=0)
for n in sizes:
    dt = torch.float64
    ts = [torch.randn(n ** 4, dtype=dt, device="cuda") for _ in range(4)]
    tB = torch.empty_like(ts[0])
    B = colmajor_view(S, tB, (n,) * 4)
    views = tuple(colmajor_view(S, t, (n,) * 4).permutedims(q) for t, q in zip(ts, perms))
    reps = 200 if n <= 32 else (30 if n <= 64 else (8 if n <= 96 else 3))
    t4 = [t.reshape((n,) * 4) for t in ts]
    cm = lambda t, p: t.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
    ref4 = (((cm(t4[0], perms[0]) + cm(t4[1], perms[1])) + cm(t4[2], perms[2])) + cm(t4[3], perms[3])).contiguous().reshape(-1)
    ref3 = (cm(t4[0], perms[0]) + cm(t4[1], perms[1]) * cm(t4[2], perms[3])).contiguous().reshape(-1)
    for name, kw in variants:
        setopt(**defaults)
        setopt(**kw)
        for what, f, vs, nby, ref in (("add4 of 4 distinct arrays", lambda a, b, c, d: a + b + c + d, views, 5, ref4),
                                      ("A .+ perm(C) .* perm'(D), 3 arrays", lambda a, b, c: a + b * c, (views[0], views[1], colmajor_view(S, ts[2], (n,) * 4).permutedims(perms[3])), 4, ref3)):
            try:
                plan = S.make_plan(f, None, None, B.size, (B,) + vs)
                us = time_plan(plan, reps)
                ok = torch.equal(tB, ref)
                d = plan.describe()
                print(f"{what:36s} {n:4d}^4 f64 | {name:32s} {us:10.2f} us {nby * 8 * n ** 4 / us / 1e3:8.1f} GB/s ({nby}N bytes) {'ok' if ok else 'WRONG'} | {d[d.find('tile='):d.find(' algb')]}")
            except Exception as e:  # noqa: BLE001
                print(f"{what:36s} {n:4d}^4 f64 | {name:32s} failed: {str(e)[:100]}")
            sys.stdout.flush()
    setopt(**defaults)
    del ts, tB, t4, ref4, ref3
    torch.cuda.empty_cache()
