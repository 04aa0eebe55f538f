#!/usr/bin/env python3
"""REDUCE_ALL with plain / non-temporal loads (option nt_load), GPU box only: configs[3] on one GPU and smaller sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

fn = S.fn


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


for dims in ((4096, 4096, 64), (4096, 4096, 8), (2048, 2048, 4)):
    numel = dims[0] * dims[1] * dims[2]
    tA = torch.rand(numel, dtype=torch.float32, device="cuda") * 2 - 1
    A = colmajor_view(S, tA, dims)
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    O = S.StridedView(out, A.size, (0, 0, 0), 0)
    row = []
    for ntl in (0, 1, 0, 1):
        S.set_option("nt_load", ntl)
        plan = S.make_plan(fn.abs2, "+", "zero", A.size, (O, A))
        plan.execute(cur())
        torch.cuda.synchronize()
        val = float(out.item())
        reps = max(3, min(50, int(2e10 / (numel * 4))))
        g = graph_of(torch, lambda: plan.execute(cur()), reps)
        g.replay()
        torch.cuda.synchronize()
        us = min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3
        row.append(f"ntl={ntl}: {us:8.2f} us {numel * 4 / us / 1e3:7.1f} GB/s (sum {val:.6e})")
    print(f"abs2-sum {dims} f32 | " + " | ".join(row))
S.set_option("nt_load", 0)
