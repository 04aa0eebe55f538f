#!/usr/bin/env python3
"""Small problems through the GENERIC family (latency-bound), GPU box only."""
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


S.set_option("force_family", 1)  # FAM_GENERIC
for dims, p in (((4,) * 8, (0, 3, 5, 4, 2, 6, 7, 1)), ((7, 9, 5, 3), (3, 2, 1, 0)), ((13, 3, 11), (2, 0, 1)), ((3,) * 5, (4, 3, 2, 1, 0)),
                ((32, 32, 32), (2, 1, 0)), ((64, 64), (1, 0)), ((100, 90, 80), (2, 1, 0))):
    n = int(np.prod(dims))
    A = colmajor_view(S, torch.randn(n, dtype=torch.float64, device="cuda"), dims)
    B = colmajor_view(S, torch.empty(n, dtype=torch.float64, device="cuda"), tuple(dims[i] for i in p))
    plan = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(p)))
    plan.execute(cur())
    torch.cuda.synchronize()
    ok = np.array_equal(B.toarray(), np.transpose(A.toarray(), p))
    g = graph_of(torch, lambda: plan.execute(cur()), 300)
    g.replay()
    torch.cuda.synchronize()
    us = min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / 300 * 1e3
    d = plan.describe()
    print(f"{str(dims):28s} {str(p):26s} {us:7.2f} us {'ok' if ok else 'WRONG'} | {d[d.find('family='):d.find(' ct=')]}")
S.set_option("force_family", 0)
