#!/usr/bin/env python3
"""Sweeps the tile-shape / launch options of the TILED family on the headline workloads and
prints a table (GPU box only).  Usage: python tools/tune.py [--n 32] [--reps 200]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3  # us


def reset():
    for i in range(8):
        S.set_option(f"tile_lg{i}", -1)
    S.set_option("tile_log2", 0)
    S.set_option("tiled_vec", 1)
    S.set_option("force_family", 0)


def compositions(total, parts, lo=1, hi=5):
    """All `parts`-tuples of log2 tile extents in [lo, hi] that sum to `total`."""
    if parts == 1:
        return [(total,)] if lo <= total <= hi else []
    return [(v,) + rest for v in range(lo, hi + 1) for rest in compositions(total - v, parts - 1, lo, hi)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--sweep", type=int, default=0, help="sweep every tile shape of 2^SWEEP elements (10 or 12) on bcast4")
    args = ap.parse_args()
    n = args.n
    dt = {"f64": torch.float64, "f32": torch.float32, "c64": torch.complex64, "c128": torch.complex128}[args.dtype]
    tA = torch.randn(n ** 4, dtype=dt, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    work = {
        "perm4321": (lambda x: x, (B, A.permutedims((3, 2, 1, 0)))),
        "perm2341": (lambda x: x, (B, A.permutedims((1, 2, 3, 0)))),
        "perm3412": (lambda x: x, (B, A.permutedims((2, 3, 0, 1)))),
        "bcast4": (lambda a, b, c, d: a + b + c + d, (B,) + tuple(A.permutedims(q) for q in perms)),
    }
    tiles = {
        "perm4321": [None],
        "perm2341": [None],
        "perm3412": [None],
        "bcast4": [None, "tl10", "tl12"] + (compositions(args.sweep, 4, 2 if args.sweep == 12 else 1) if args.sweep else []),
    }
    elem = tA.element_size()
    algb = 2 * elem * n ** 4
    print(f"# n={n} dtype={args.dtype} algorithmic bytes/launch={algb}")
    # floor: plain contiguous copy of the same bytes (STREAM family) and torch's own copy kernel
    reset()
    plan = S.make_plan(lambda x: x, None, None, A.size, (B, A))
    us = time_plan(plan, args.reps)
    print(f"copy      STREAM          {us:9.2f} us {algb / us / 1e3:8.1f} GB/s  | {plan.describe()}")
    g = graph_of(torch, lambda: tB.copy_(tA), args.reps)
    g.replay()
    torch.cuda.synchronize()
    us = min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / args.reps * 1e3
    print(f"copy      torch.copy_     {us:9.2f} us {algb / us / 1e3:8.1f} GB/s")
    sys.stdout.flush()
    for name, (f, arrays) in work.items():
        for tile in tiles[name]:
            for vec in (1,):
                reset()
                S.set_option("max_lds_bytes", 160 * 1024)
                if tile in ("tl10", "tl12"):
                    S.set_option("tile_log2", int(tile[2:]))
                elif tile is not None:
                    if sum(tile) == 12:
                        S.set_option("tile_log2", 12)
                    for i, v in enumerate(tile):
                        S.set_option(f"tile_lg{i}", v)
                S.set_option("tiled_vec", vec)
                try:
                    plan = S.make_plan(f, None, None, A.size, arrays)
                    us = time_plan(plan, args.reps)
                    d = plan.describe()
                    print(f"{name:9s} tile={str(tile):16s} vec={vec} {us:9.2f} us {algb / us / 1e3:8.1f} GB/s  | "
                          f"{d[d.find('tile='):d.find(' algbytes')] if 'tile=' in d else d}")
                except Exception as e:  # noqa: BLE001
                    print(f"{name:9s} tile={tile} vec={vec}: {type(e).__name__}: {e}")
                sys.stdout.flush()
    # generic-kernel reference point
    reset()
    S.set_option("force_family", 1)
    for name, (f, arrays) in work.items():
        plan = S.make_plan(f, None, None, A.size, arrays)
        us = time_plan(plan, max(5, args.reps // 10))
        print(f"{name:9s} GENERIC fallback {us:9.2f} us {algb / us / 1e3:8.1f} GB/s")
    reset()


if __name__ == "__main__":
    main()
