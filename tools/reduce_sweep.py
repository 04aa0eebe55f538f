#!/usr/bin/env python3
"""Complete reduction mapreduce(abs2, +, A) on 4 GiB of Float32 (config 4): workgroup-count sweep
against torch's own reductions on the same buffer (GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms  # noqa: E402


def main():
    n = 4096 * 4096 * 64
    t = torch.randn(n, dtype=torch.float32, device="cuda")
    A = colmajor_view(S, t, (4096, 4096, 64))
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    o = colmajor_view(S, out, (1,))
    s = int(torch.cuda.current_stream().cuda_stream)
    nbytes = 4 * n
    for name, fn in (("torch.sum", lambda: t.sum()), ("torch.dot(t,t)", lambda: torch.dot(t, t)), ("torch.linalg.vector_norm", lambda: torch.linalg.vector_norm(t))):
        fn()
        torch.cuda.synchronize()
        ms = min(event_time_ms(torch, fn, 3) for _ in range(3))
        print(f"{name:28s} {ms * 1e3:9.1f} us {nbytes / ms / 1e6:8.1f} GB/s")
    for nb in (512, 1024, 2048, 4096, 8192, 16384, 65536):
        S.set_option("reduce_blocks", nb)
        plan = S.make_plan(S.fn.abs2, "+", None, A.size, S.promoteshape(A.size, o.sreshape((1, 1, 1)), A))
        plan.execute(s)
        torch.cuda.synchronize()
        ms = min(event_time_ms(torch, lambda: plan.execute(s), 3) for _ in range(3))
        print(f"smr reduce_all blocks={nb:6d} {ms * 1e3:9.1f} us {nbytes / ms / 1e6:8.1f} GB/s | {plan.describe()}")
        sys.stdout.flush()
    S.set_option("reduce_blocks", 2048)


if __name__ == "__main__":
    main()
