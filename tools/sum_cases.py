#!/usr/bin/env python3
"""Partial sums of (100, 90, 80, 7), Float32 and Float64, every dim subset: us per launch (hipGraph, HIP events) and GB/s.
Usage: [SMR_LIB=...] python tools/sum_cases.py [opt=value ...]"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

lib = S._lib.load()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    S._lib.check(lib.smr_set_option(k.encode(), int(v)))


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps=40):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3


print("library:", os.environ.get("SMR_LIB", "product"), " ".join(sys.argv[1:]))
for dims, dt in (((100, 90, 80, 7), torch.float32), ((100, 90, 80, 7), torch.float64), ((512, 384, 64), torch.float32)):
    n = int(np.prod(dims))
    A = colmajor_view(S, torch.randn(n, dtype=dt, device="cuda"), dims)
    ref = torch.as_strided(A.parent if hasattr(A, "parent") else A.base, dims, [int(np.prod(dims[:i])) for i in range(len(dims))]) if False else None
    for k in range(1, len(dims)):
        for rd in itertools.combinations(range(len(dims)), k):
            out = A.similar(size=tuple(1 if d in rd else m for d, m in enumerate(dims)))
            plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
            d = plan.describe()
            us = time_plan(plan)
            print("%-8s %-18s dims=%-10s %7.2f us %6.0f GB/s | %s" % (str(dt)[6:], dims, rd, us, plan.algorithmic_bytes / us / 1e3, d[d.find("dims="):d.find(" algbytes")]))
            sys.stdout.flush()
