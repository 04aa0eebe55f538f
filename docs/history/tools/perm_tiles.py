#!/usr/bin/env python3
"""Tile-shape / persistence sweep for big transposing copies through the classic TILED kernel (GPU box only):
permutedims!(B, A, (4,3,2,1)) at 128^4 f64 and the 8192^2 / 16384^2 transposes."""
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
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3


def reset():
    for i in range(8):
        S.set_option(f"tile_lg{i}", -1)
    S.set_option("tile_log2", 0)
    S.set_option("tiled_persist", 1)
    S.set_option("tiled_persist_wpc", 0)
    S.set_option("max_lds_bytes", 65536)


def main():
    n = 128
    tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
    jobs = [("perm4321 128^4 f64", (B, A.permutedims((3, 2, 1, 0))), (0, 3), 16 * n ** 4)]
    for name, arrays, axes, algb in jobs:
        for lg in ((5, 5), (6, 4), (4, 6), (6, 6), (7, 5), (5, 7), (7, 3), (3, 7)):
            for persist, wpc in ((0, 0), (1, 0), (1, 2), (1, 8)):
                reset()
                S.set_option("max_lds_bytes", 160 * 1024)
                if sum(lg) == 12:
                    S.set_option("tile_log2", 12)
                S.set_option(f"tile_lg{axes[0]}", lg[0])
                S.set_option(f"tile_lg{axes[1]}", lg[1])
                for d in range(4):
                    if d not in axes:
                        S.set_option(f"tile_lg{d}", 0)
                S.set_option("tiled_persist", persist)
                S.set_option("tiled_persist_wpc", wpc)
                try:
                    plan = S.make_plan(lambda x: x, None, None, arrays[0].size, arrays)
                    d = plan.describe()
                    if "family=tiled" not in d:
                        print(f"{name} tile={lg} -> {d[:60]}")
                        break
                    us = time_plan(plan, 4)
                    print(f"{name} tile={lg} persist={persist} wpc={wpc} {us:9.1f} us {algb / us / 1e3:8.1f} GB/s | {d[d.find('tile='):d.find(' algb')]}")
                except Exception as e:  # noqa: BLE001
                    print(f"{name} tile={lg}: {type(e).__name__}: {str(e)[:100]}")
                    break
                sys.stdout.flush()
    del tA, tB
    for m in (8192, 16384):
        tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        for lg in ((5, 5), (6, 6), (6, 4), (4, 6), (7, 5), (5, 7)):
            for persist in (0, 1):
                reset()
                S.set_option("max_lds_bytes", 160 * 1024)
                if sum(lg) == 12:
                    S.set_option("tile_log2", 12)
                S.set_option("tile_lg0", lg[0])
                S.set_option("tile_lg1", lg[1])
                S.set_option("tiled_persist", persist)
                try:
                    plan = S.make_plan(lambda x: x, None, None, (m, m), (B, A.permutedims((1, 0))))
                    d = plan.describe()
                    us = time_plan(plan, 6)
                    print(f"transpose {m}^2 tile={lg} persist={persist} {us:9.1f} us {16 * m * m / us / 1e3:8.1f} GB/s | {d[d.find('tile='):d.find(' algb')]}")
                except Exception as e:  # noqa: BLE001
                    print(f"transpose {m}^2 tile={lg}: {type(e).__name__}: {str(e)[:100]}")
                sys.stdout.flush()
        del tA, tB
    reset()


if __name__ == "__main__":
    main()
