#!/usr/bin/env python3
"""Experiment: vectors in flight per lane in the STREAM family (option stream_u; runtime-compiled functors)."""
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


def main():
    for dt, m in ((torch.float32, 8192), (torch.float64, 8192)):
        ts = [torch.rand(m * m, dtype=dt, device="cuda") + 0.5 for _ in range(3)]
        B, A, C = (colmajor_view(S, t, (m, m)) for t in ts)
        for label, f, arrs in (("a*exp(-3a)+cos(a*a)", lambda a: a * fn.exp(-3 * a) + fn.cos(a * a), (B, A)),
                               ("a*2+c/3-1", lambda a, c: a * 2 + c / 3 - 1, (B, A, C)),
                               ("a-0.25 (copy-like)", lambda a: a - 0.25, (B, A))):
            for u in (0, 2, 8):
                S.set_option("stream_u", u)
                plan = S.make_plan(f, None, None, arrs[0].size, arrs)
                plan.execute(cur())
                torch.cuda.synchronize()
                g = graph_of(torch, lambda: plan.execute(cur()), 10)
                g.replay()
                torch.cuda.synchronize()
                us = min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / 10 * 1e3
                print(f"{str(dt)[6:]:8s} {label:22s} U={u or 'auto(4)'}: {us:8.1f} us {plan.algorithmic_bytes / us / 1e3:8.1f} GB/s")
    S.set_option("stream_u", 0)


if __name__ == "__main__":
    main()
