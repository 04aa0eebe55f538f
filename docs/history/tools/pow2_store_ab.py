#!/usr/bin/env python3
"""Round 5: store policy at HBM sizes.  4-way permuted sum and permutedims!(4,3,2,1) at 96^4 ... 144^4 Float64 with nt_store = auto /
off / on, in the product build and in the sc1 experiment build (nt stores -> agent-scope write-through, `global_store ... sc1`).
HIP events over graph-replayed launches (as bench.py's secondary()); GB/s of algorithmic bytes (2 x 8 x n^4)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SC1 = os.path.join(ROOT, "strided.jl_amd", "libstrided_hip_sc1.so")
if "--child" not in sys.argv:
    for name, lib in (("product build", None), ("sc1 build", SC1)):
        if lib and not os.path.exists(lib):
            continue
        env = dict(os.environ)
        if lib:
            env["SMR_LIB"] = lib
        print("==== %s ====" % name, flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child"] + [a for a in sys.argv[1:]], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [96, 120, 128, 136, 144]
dev = torch.device("cuda", 0)
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def timed(plan, reps=5):
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(3)) / reps


for n in sizes:
    tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
    row = []
    for nt in (-1, 0, 1):
        S.set_option("nt_store", nt)
        p = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
        ms3 = timed(p)
        d3 = p.describe()
        p = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
        ms2 = timed(p)
        row.append("nt=%2d: sum4 %8.1f us %6.0f GB/s | perm %8.1f us %6.0f GB/s" % (nt, ms3 * 1e3, 16 * n ** 4 / ms3 / 1e6, ms2 * 1e3, 16 * n ** 4 / ms2 / 1e6))
    print("%4d^4  %s" % (n, "  ||  ".join(row)), flush=True)
    del tA, tB
S.set_option("nt_store", -1)
