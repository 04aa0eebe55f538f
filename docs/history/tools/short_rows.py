#!/usr/bin/env python3
"""Maps over views whose contiguous rows are short (sub-boxes): STREAM family utilisation (GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def main():
    for rows, full in ((100, 128), (24, 32), (500, 512), (8, 8)):
        t = torch.randn(full * 1000 * 500, dtype=torch.float32, device="cuda")
        A = colmajor_view(S, t, (full, 1000, 500))
        V = A.sview(slice(0, rows), slice(None), slice(None)) if rows != full else A.sview(slice(None), slice(0, 999), slice(None))
        o = torch.empty(V.size[0] * V.size[1] * V.size[2], dtype=torch.float32, device="cuda")
        O = colmajor_view(S, o, V.size)
        plan = S.make_plan(lambda x: x * 2, None, None, V.size, (O, V))
        plan.execute(cur())
        torch.cuda.synchronize()
        ms = min(event_time_ms(torch, lambda: plan.execute(cur()), 3) for _ in range(3))
        n = V.size[0] * V.size[1] * V.size[2]
        print(f"rows of {V.size[0]:4d} in {full:4d} x {V.size[1]} x {V.size[2]}: {ms * 1e3:9.1f} us {8 * n / ms / 1e6:8.1f} GB/s | {plan.describe()}")


if __name__ == "__main__":
    main()
