#!/usr/bin/env python3
"""Round 5: STREAM rows that are not whole aligned vectors -- element-aligned 16-byte accesses + one partial vector per row (option
stream_ua = 1) against 4- / 8-byte accesses (0).  permutedims with an unchanged unit axis, axpy-like maps on odd shapes; HIP events
over graph-replayed launches; every result checked against torch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

dev = torch.device("cuda", 0)


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def timed(plan, reps=50):
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(5)) / reps


def case(shape, perm, dtype, nary=False):
    n = 1
    for d in shape:
        n *= d
    tA = torch.randn(n, dtype=dtype, device=dev)
    tC = torch.randn(n, dtype=dtype, device=dev)
    tB = torch.empty_like(tA)
    rank = len(shape)
    oshape = tuple(shape[p] for p in perm)
    A, B, Cc = colmajor_view(S, tA, shape), colmajor_view(S, tB, oshape), colmajor_view(S, tC, oshape)
    at = tA.reshape(tuple(reversed(shape))).permute(*[rank - 1 - perm[rank - 1 - i] for i in range(rank)]).contiguous().reshape(-1)
    want = at * 2 + tC if nary else at
    row = []
    for ua in (0, 1):
        S.set_option("stream_ua", ua)
        if nary:
            p = S.make_plan(lambda x, y: x * 2 + y, None, None, B.size, (B, A.permutedims(perm), Cc))
        else:
            p = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(perm)))
        tB.zero_()
        p.execute(cur())
        torch.cuda.synchronize()
        ok = torch.equal(tB, want)
        ms = timed(p)
        row.append("%s %7.2f us %5.0f GB/s%s" % ("vectors+tail" if ua else "scalar      ", ms * 1e3, (3 if nary else 2) * tA.element_size() * n / ms / 1e6, "" if ok else " WRONG"))
        d = p.describe()
    S.set_option("stream_ua", 1)
    print("%-20s %-12s %-8s %-5s | %s | %s" % (shape, perm, str(dtype).split(".")[-1], "n-ary" if nary else "copy", " | ".join(row), d[:100]), flush=True)


case((257, 129, 65), (0, 2, 1), torch.float64)
case((257, 129, 65), (0, 2, 1), torch.float64, True)
case((17, 33, 65, 31), (0, 2, 1, 3), torch.float64)
case((17, 33, 65, 31), (0, 1, 3, 2), torch.float64)
case((999, 1001), (0, 1), torch.float64, True)
case((1001, 999, 5), (0, 2, 1), torch.float32)
case((1001, 999, 5), (0, 2, 1), torch.float32, True)
case((301, 303, 35), (0, 2, 1), torch.float64)
case((63, 500, 500), (0, 2, 1), torch.float32)
case((256, 129, 65), (0, 2, 1), torch.float64)
