#!/usr/bin/env python3
"""Partial / complete reductions by op (+, max, min, *), GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


for dims, rd in (((512, 384, 64), (0,)), ((512, 384, 64), (2,)), ((100, 90, 80, 7), (0, 1)), ((100, 90, 80, 7), (1, 3)), ((4096, 4096), (0, 1))):
    n = int(np.prod(dims))
    A = colmajor_view(S, torch.rand(n, dtype=torch.float32, device="cuda") + 0.5, dims)
    out = A.similar(size=tuple(1 if d in rd else m for d, m in enumerate(dims)))
    row = []
    for op in ("+", "max", "min", "*"):
        plan = S.make_plan(lambda x: x, op, "identity" if op != "+" else "zero", dims, S.promoteshape(dims, out, A))
        plan.execute(cur())
        torch.cuda.synchronize()
        g = graph_of(torch, lambda: plan.execute(cur()), 50)
        g.replay()
        torch.cuda.synchronize()
        us = min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / 50 * 1e3
        row.append(f"{op}: {us:7.2f} us")
    print(f"{str(dims):20s} dims={str(rd):10s} | " + " | ".join(row))
