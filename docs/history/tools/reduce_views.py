#!/usr/bin/env python3
"""Complete reductions over views that do not fuse into one contiguous run (GPU box only)."""
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
    t = torch.randn(1024 * 1024 * 256, dtype=torch.float32, device="cuda")
    A = colmajor_view(S, t, (1024, 1024, 256))
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    o = colmajor_view(S, out, (1,))
    tv = t.view(256, 1024, 1024)
    cases = [("whole (fuses to 1-d)", A, lambda: tv.sum()),
             ("A[0:1000, 0:1000, :] (box)", A.sview(slice(0, 1000), slice(0, 1000), slice(None)), lambda: tv[:, 0:1000, 0:1000].sum()),
             ("A[::2, :, :] (stride 2)", A.sview(slice(0, 1024, 2), slice(None), slice(None)), lambda: tv[:, :, ::2].sum()),
             ("permuted box", A.sview(slice(0, 1000), slice(0, 1000), slice(None)).permutedims((2, 0, 1)), lambda: tv[:, 0:1000, 0:1000].sum())]
    for name, V, tf in cases:
        plan = S.make_plan(lambda x: x, "+", None, V.size, S.promoteshape(V.size, o.sreshape((1,) * V.ndim), V))
        plan.execute(cur())
        torch.cuda.synchronize()
        ms = min(event_time_ms(torch, lambda: plan.execute(cur()), 3) for _ in range(3))
        n = 1
        for d in V.size:
            n *= d
        tf()
        torch.cuda.synchronize()
        mst = min(event_time_ms(torch, tf, 3) for _ in range(3))
        got = float(out.item())
        print(f"{name:30s} smr {ms * 1e3:9.1f} us {4 * n / ms / 1e6:8.1f} GB/s | torch {mst * 1e3:9.1f} us | {plan.describe()}")
        out.zero_()


if __name__ == "__main__":
    main()
