#!/usr/bin/env python3
"""Tile-edge sweep of the ORBIT family (option orbit_lg) against the classic tiled kernel over problem
sizes, for the 4-way permuted sum and the symmetrise (GPU box only)."""
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


def main():
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    for dt in (torch.float64, torch.float32):
        for n in (16, 24, 32, 40, 48, 64, 80, 96, 128):
            if dt == torch.float32 and n > 96:
                continue
            tA = torch.randn(n ** 4, dtype=dt, device="cuda")
            tB = torch.empty_like(tA)
            A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
            reps = max(3, min(200, int(2e8 / n ** 4)))
            row = []
            for name, opts in (("classic", dict(orbit=0)), ("lg2", dict(orbit=1, orbit_lg=2)), ("lg3", dict(orbit=1, orbit_lg=3)), ("auto", dict(orbit=1, orbit_lg=-1))):
                for k, v in opts.items():
                    S.set_option(k, v)
                plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
                d = plan.describe()
                fam = d[d.find("family=") + 7:d.find(" ct=")]
                if name.startswith("lg") and fam != "orbit":
                    row.append(f"{name}: -")
                    continue
                us = time_plan(plan, reps)
                row.append(f"{name}[{fam}{' ' + d[d.find('tile='):d.find(' group')] if fam == 'orbit' else ''}]: {us:8.2f} us {2 * tA.element_size() * n ** 4 / us / 1e3:7.1f} GB/s")
            print(f"bcast4 {n}^4 {str(dt)[6:]:8s} | " + " | ".join(row))
            sys.stdout.flush()
            del tA, tB
    S.set_option("orbit_lg", -1)
    for m in (512, 1024, 2048, 4000, 4096, 8192, 12000, 16384):
        tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        reps = max(3, min(200, int(4e8 / (m * m))))
        row = []
        for name, opts in (("classic", dict(orbit=0)), ("lg4", dict(orbit=1, orbit_lg=4)), ("lg5", dict(orbit=1, orbit_lg=5)), ("lg6", dict(orbit=1, orbit_lg=6)), ("auto", dict(orbit=1, orbit_lg=-1))):
            for k, v in opts.items():
                S.set_option(k, v)
            plan = S.make_plan(lambda x, y: (x + y) / 2, None, None, (m, m), (B, A, A.permutedims((1, 0))))
            d = plan.describe()
            fam = d[d.find("family=") + 7:d.find(" ct=")]
            if name.startswith("lg") and fam != "orbit":
                row.append(f"{name}: -")
                continue
            us = time_plan(plan, reps)
            row.append(f"{name}[{fam}]: {us:8.2f} us {16 * m * m / us / 1e3:7.1f} GB/s")
        print(f"sym {m}^2 f64 | " + " | ".join(row))
        sys.stdout.flush()
        del tA, tB
    S.set_option("orbit_lg", -1)
    S.set_option("orbit", 1)


if __name__ == "__main__":
    main()
