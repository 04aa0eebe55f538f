#!/usr/bin/env python3
"""permutedims! / transposes / distinct-array permuted sums through the classic TILED kernel (GPU box
only).  Run twice with SMR_LIB pointing at different builds for A/B (e.g. -DSMR_TILED_BITS=0)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(4)) / reps * 1e3


def main():
    tag = os.environ.get("TAG", "")
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    for dt in (torch.float64, torch.float32, torch.complex128):
        for n in (32, 64, 128):
            if dt == torch.complex128 and n > 64:
                continue
            tA = torch.randn(n ** 4, dtype=dt, device="cuda")
            tB = torch.empty_like(tA)
            A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
            reps = 300 if n <= 32 else (40 if n <= 64 else 4)
            for name, q in (("perm4321", (3, 2, 1, 0)), ("perm2341", (1, 2, 3, 0)), ("perm3412", (2, 3, 0, 1))):
                plan = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims(q)))
                us = time_plan(plan, reps)
                ok = torch.equal(tB.reshape((n,) * 4), tA.reshape((n,) * 4).permute(*[3 - q[3 - i] for i in range(4)]).contiguous())
                print(f"{tag} {name} {n}^4 {str(dt)[6:]:10s} {us:9.2f} us {2 * tA.element_size() * n ** 4 / us / 1e3:8.1f} GB/s {'ok' if ok else 'WRONG'}")
            if dt == torch.float64 and n <= 64:
                others = [torch.randn(n ** 4, dtype=dt, device="cuda") for _ in range(3)]
                views = (A,) + tuple(colmajor_view(S, t, (n,) * 4).permutedims(q) for t, q in zip(others, perms[1:]))
                plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + views)
                us = time_plan(plan, reps)
                d = plan.describe()
                print(f"{tag} add4 of 4 distinct arrays {n}^4 f64 {us:9.2f} us {5 * 8 * n ** 4 / us / 1e3:8.1f} GB/s (5N bytes) | {d[d.find('family='):d.find(' ct=')]} {d[d.find('tile='):d.find(' algb')]}")
                del others
            sys.stdout.flush()
            del tA, tB
    for m in (4000, 8192):
        tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        plan = S.make_plan(lambda x: x, None, None, (m, m), (B, A.permutedims((1, 0))))
        us = time_plan(plan, 20)
        print(f"{tag} transpose {m}^2 f64 {us:9.2f} us {16 * m * m / us / 1e3:8.1f} GB/s")
        plan = S.make_plan(lambda x: 3 * x, None, None, (m, m), (B, A.permutedims((1, 0))))
        us = time_plan(plan, 20)
        print(f"{tag} 3 .* A' {m}^2 f64 {us:9.2f} us {16 * m * m / us / 1e3:8.1f} GB/s")
        del tA, tB


if __name__ == "__main__":
    main()
