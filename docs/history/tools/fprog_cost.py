#!/usr/bin/env python3
"""How much does the f-program interpreter cost against natively compiled functors?  (GPU box only)"""
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


def time_plan(plan, reps=20):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3


def main():
    for dt, name in ((torch.float32, "f32"), (torch.float64, "f64"), (torch.complex64, "c32")):
        m = 8192 if dt != torch.float64 else 4096
        ts = [torch.rand(m * m, dtype=dt, device="cuda") + 0.5 for _ in range(4)]
        B, A, C, D = (colmajor_view(S, t, (m, m)) for t in ts)
        rows = [
            ("copy (native ident)", lambda a: a, (B, A)),
            ("a+c (native add2)", lambda a, c: a + c, (B, A, C)),
            ("a-c (interpreted, 3 instr)", lambda a, c: a - c, (B, A, C)),
            ("a*2+c/3-1 (interpreted, 9 instr)", lambda a, c: a * 2 + c / 3 - 1, (B, A, C)),
            ("a*c+d*a-c*d+a (interpreted, 13 instr)", lambda a, c, d: a * c + d * a - c * d + a, (B, A, C, D)),
            ("sqrt(abs(a))*c (interpreted)", lambda a, c: fn.sqrt(fn.abs(a)) * c, (B, A, C)),
            ("a*exp(-2a)+sin(a*a) (native expr5)", lambda a: a * fn.exp(-2 * a) + fn.sin(a * a), (B, A)),
            ("a*exp(-3a)+cos(a*a) (interpreted)", lambda a: a * fn.exp(-3 * a) + fn.cos(a * a), (B, A)),
            ("transposed: a' - c (interpreted, tiled)", lambda a, c: a - c, (B, A.permutedims((1, 0)), C)),
            ("transposed: a' + c (native, tiled)", lambda a, c: a + c, (B, A.permutedims((1, 0)), C)),
        ]
        for label, f, arrs in rows:
            try:
                p = S.make_plan(f, None, None, arrs[0].size, arrs)
            except Exception as e:  # noqa: BLE001
                print(f"{name} {label:42s} {type(e).__name__}: {e}")
                continue
            us = time_plan(p)
            d = p.describe()
            print(f"{name} {label:42s} {us:9.1f} us {p.algorithmic_bytes / us / 1e3:8.1f} GB/s | {d[:d.find(' N=')]}")
            sys.stdout.flush()


if __name__ == "__main__":
    main()
