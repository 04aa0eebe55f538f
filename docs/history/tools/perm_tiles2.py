#!/usr/bin/env python3
"""4096-element tile shapes for big two-axis transposing copies (TILED kernel, non-temporal stores on), GPU box only.
dst unit axis first: (a, b) = 2^a along the destination's unit axis x 2^b along the source's."""
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
    jobs = []
    for dt in (torch.float64, torch.float32, torch.complex128):
        for kind, n in (("perm4321", 64), ("perm4321", 128), ("transpose", 4000), ("transpose", 8192), ("transpose", 16384), ("transpose", 12000)):
            if dt == torch.complex128 and (n == 128 or n == 16384):
                continue
            jobs.append((kind, n, dt))
    for kind, n, dt in jobs:
        numel = n ** 4 if kind == "perm4321" else n * n
        dims = (n,) * 4 if kind == "perm4321" else (n, n)
        tA = torch.randn(numel, dtype=dt, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, dims), colmajor_view(S, tB, dims)
        arrays = (B, A.permutedims(tuple(reversed(range(len(dims))))))
        axes = (0, len(dims) - 1)
        reps = max(3, min(40, int(2e9 / (numel * tA.element_size()))))
        row = []
        reset()
        plan = S.make_plan(lambda x: x, None, None, dims, arrays)
        us = time_plan(plan, reps)
        d = plan.describe()
        row.append(f"auto[{d[d.find('tile=') + 5:d.find(' staged')]}] {us:8.1f}")
        for lg in ((5, 5), (6, 6), (7, 5), (8, 4), (5, 7)):
            for persist in (0, 1):
                reset()
                S.set_option("max_lds_bytes", 160 * 1024)
                if sum(lg) == 12:
                    S.set_option("tile_log2", 12)
                for dd in range(len(dims)):
                    S.set_option(f"tile_lg{dd}", 0)
                S.set_option(f"tile_lg{axes[0]}", lg[0])
                S.set_option(f"tile_lg{axes[1]}", lg[1])
                S.set_option("tiled_persist", persist)
                try:
                    plan = S.make_plan(lambda x: x, None, None, dims, arrays)
                    if "family=tiled" not in plan.describe():
                        row.append(f"{lg}p{persist} -")
                        continue
                    us = time_plan(plan, reps)
                    row.append(f"{lg[0]},{lg[1]}p{persist} {us:8.1f}")
                except Exception as e:  # noqa: BLE001
                    row.append(f"{lg}p{persist} {type(e).__name__}")
        print(f"{kind} {n} {str(dt)[6:]:10s} ({2 * numel * tA.element_size() >> 20} MiB) | " + " | ".join(row))
        sys.stdout.flush()
        del tA, tB
    reset()


if __name__ == "__main__":
    main()
