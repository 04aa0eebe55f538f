#!/usr/bin/env python3
"""Graph-replayed launches of the two headline kernels ONLY (for a rocprofv3 kernel trace whose
per-dispatch durations can be compared with bench.py's HIP-event numbers): each plan is executed
eagerly exactly `--eager` times (table upload + capture warm-up), then `--replays` replays of a
hipGraph holding `--reps` back-to-back launches of ONE kernel.
Usage (GPU box): rocprofv3 --kernel-trace -d out -o kt -- python tools/graph_trace.py
                 python tools/rocpd_summary.py --hist out/kt_results.db"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--reps", type=int, default=500)
ap.add_argument("--replays", type=int, default=8)
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
}


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


for name, plan in plans.items():
    print(name, plan.describe())
    g = graph_of(torch, lambda: plan.execute(cur()), args.reps)  # 1 eager launch + the captured ones
    g.replay()
    torch.cuda.synchronize()
    ms = [event_time_ms(torch, g.replay, 1) / args.reps for _ in range(args.replays)]
    print(f"{name}: HIP events over graph replays: min {min(ms) * 1e3:.3f} us, median {sorted(ms)[len(ms) // 2] * 1e3:.3f} us per launch "
          f"({args.replays} replays x {args.reps} launches; under rocprofv3 these include the profiler's per-dispatch overhead)")
torch.cuda.synchronize()
