#!/usr/bin/env python3
"""Round 5: option sweep of the 4-way permuted sum and permutedims! at 128^4 Float64 (the power-of-two collapse).
HIP events over graph-replayed launches; every option combination is a fresh plan."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
what = sys.argv[2] if len(sys.argv) > 2 else "sum"
dev = torch.device("cuda", 0)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
tB = torch.empty_like(tA)
A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def timed(plan, reps=4):
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(3)) / reps


def run(opts):
    saved = {k: S.get_option(k) for k in opts}
    for k, v in opts.items():
        S.set_option(k, v)
    try:
        if what == "sum":
            p = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
        else:
            p = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
        ms = timed(p)
        d = p.describe()
    except Exception as e:  # noqa: BLE001
        ms, d = float("nan"), "ERR " + str(e)[:80]
    for k, v in saved.items():
        S.set_option(k, v)
    print("%-70s %9.1f us %6.0f GB/s | %s" % (" ".join("%s=%s" % kv for kv in opts.items()) or "(default)", ms * 1e3, 16 * n ** 4 / ms / 1e6, d[:110]), flush=True)


run({})
if what == "sum":
    run({"orbit_deal": 1})
    run({"orbit_deal": 1, "orbit_pipe": 0})
    run({"orbit_deal": 1, "orbit_group": 4})
    run({"orbit_deal": 1, "orbit_group": 1})
    run({"orbit_deal": 1, "nt_store": 2})
    if len(sys.argv) > 3:
        sys.exit(0)
    for g in (1, 4):
        run({"orbit_group": g})
    for sk in (1, 3, 5):
        run({"orbit_skew": sk})
    for pipe in (0, 1):
        run({"orbit_pipe": pipe})
    for w in (256, 512, 1024):
        run({"orbit_pipe": 1, "orbit_wgs": w})
    for nt in (0, 1, 2):
        run({"nt_store": nt})
    run({"orbit_pipe": 0, "orbit_lds_min": 160 * 1024})
    run({"orbit": 0})
    run({"orbit": 0, "tile_log2": 12})
    run({"orbit": 0, "tile_block": 4})
else:
    for nt in (0, 1, 2):
        run({"nt_store": nt})
    for tl in (10, 12):
        run({"tile_log2": tl})
    run({"tiled_persist": 0})
    run({"tiled_persist": 1})
    run({"force_family": 6})
