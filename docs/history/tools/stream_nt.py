#!/usr/bin/env python3
"""STREAM family with / without non-temporal stores over sizes (GPU box only): configs[4], a 2-input
arithmetic map, a plain copy; every result checked against torch first."""
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
    for m in (2048, 4096, 8192, 16384):
        tA = torch.rand(m * m, dtype=torch.float32, device="cuda")
        tC = torch.rand(m * m, dtype=torch.float32, device="cuda")
        tB = torch.empty_like(tA)
        A, B, C = (colmajor_view(S, t, (m, m)) for t in (tA, tB, tC))
        for label, f, arrs, ref in (("expr5", lambda a: a * fn.exp(-2 * a) + fn.sin(a * a), (B, A), None),
                                    ("a*2+c", lambda a, c: a * 2 + c, (B, A, C), lambda: tA * 2 + tC),
                                    ("copy", lambda a: a, (B, A), lambda: tA)):
            row = []
            for nts in (0, 1, -1):
                S.set_option("nt_store", nts)
                plan = S.make_plan(f, None, None, (m, m), arrs)
                tB.zero_()
                plan.execute(cur())
                torch.cuda.synchronize()
                ok = True if ref is None else torch.equal(tB, ref())
                reps = max(3, min(100, int(3e9 / (m * m * 8))))
                g = graph_of(torch, lambda: plan.execute(cur()), reps)
                g.replay()
                torch.cuda.synchronize()
                us = min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3
                row.append(f"nts={nts}: {us:9.2f} us {plan.algorithmic_bytes / us / 1e3:7.1f} GB/s{'' if ok else ' WRONG'}")
            print(f"{label:6s} {m}^2 f32 ({plan.algorithmic_bytes >> 20} MiB) | " + " | ".join(row))
            sys.stdout.flush()
        del tA, tB, tC
    S.set_option("nt_store", -1)


if __name__ == "__main__":
    main()
