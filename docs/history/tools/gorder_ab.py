#!/usr/bin/env python3
"""Round 5: order of the TILED grid dims for HBM-sized transposing copies (option tiled_gorder: 0 canonical / 1 the input's split unit
axis second).  permutedims! in the three benchmark permutations at 96^4 ... 144^4 Float64, 2-D transposes, Float32 / ComplexF64 at
128^4; every plan checked bit for bit against torch first; HIP events over graph-replayed launches."""
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


def timed(plan, reps=4):
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(3)) / reps


def case(shape, perm, dtype):
    n = 1
    for d in shape:
        n *= d
    tA = torch.randn(n, dtype=torch.float32, device=dev).to(dtype) if dtype != torch.complex128 else torch.randn(n, dtype=torch.complex128, device=dev)
    oshape = tuple(shape[p] for p in perm)
    tB = torch.empty_like(tA)
    A, B = colmajor_view(S, tA, shape), colmajor_view(S, tB, oshape)
    rank = len(shape)
    # column-major (i1..iN) <-> torch row-major view with reversed index order
    a_t = tA.reshape(tuple(reversed(shape)))
    want = a_t.permute(*[rank - 1 - perm[rank - 1 - i] for i in range(rank)]).contiguous().reshape(-1)
    row = []
    for mode in (0, 1, -1, 9):
        S.set_option("tiled_gorder", -1 if mode == 9 else mode)
        S.set_option("tiled_xpose", 1 if mode == 9 else 0)
        p = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(perm)))
        tB.zero_()
        p.execute(cur())
        torch.cuda.synchronize()
        ok = torch.equal(tB, want)
        ms = timed(p)
        row.append("%s %8.1f us %5.0f GB/s%s" % ({0: "canonical", 1: "input-axis 2nd", -1: "auto", 9: "auto+lean kernel"}[mode], ms * 1e3, 2 * tA.element_size() * n / ms / 1e6, "" if ok else " WRONG"))
        desc = p.describe()
    S.set_option("tiled_gorder", -1)
    S.set_option("tiled_xpose", 1)
    print("%-22s %-12s %-10s | %s | %s" % (shape, perm, str(dtype).split(".")[-1], " | ".join(row), desc[:90]), flush=True)
    del tA, tB


for n in (96, 128, 144):
    for perm in ((3, 2, 1, 0), (1, 2, 3, 0), (2, 3, 0, 1)):
        case((n,) * 4, perm, torch.float64)
case((16384, 16384), (1, 0), torch.float64)
case((8192, 8192), (1, 0), torch.float64)
case((128,) * 4, (3, 2, 1, 0), torch.float32)
case((128,) * 4, (3, 2, 1, 0), torch.complex128)
case((256, 128, 128, 64), (3, 2, 1, 0), torch.float64)
case((512, 512, 512), (2, 1, 0), torch.float64)
case((512, 512, 512), (1, 0, 2), torch.float64)
case((64, 64, 64, 64, 16), (4, 3, 2, 1, 0), torch.float64)
