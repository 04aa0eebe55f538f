#!/usr/bin/env python3
"""TILED / ORBIT kernels with plain vs non-temporal stores (option nt_store 0 / 1), GPU box only."""
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
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    cases = []
    for n in (32, 64, 128):
        tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        reps = 300 if n <= 32 else (40 if n <= 64 else 4)
        cases.append((f"perm4321 {n}^4 f64", lambda x: x, A.size, (B, A.permutedims((3, 2, 1, 0))), reps, (tA, tB)))
        cases.append((f"sum4 {n}^4 f64", lambda a, b, c, d: a + b + c + d, A.size, (B,) + tuple(A.permutedims(p) for p in perms), reps, (tA, tB)))
    for n in (32, 64):
        ts = [torch.randn(n ** 4, dtype=torch.float64, device="cuda") for _ in range(5)]
        vs = [colmajor_view(S, t, (n,) * 4) for t in ts]
        cases.append((f"add4 distinct {n}^4 f64", lambda a, b, c, d: a + b + c + d, vs[0].size,
                      (vs[1], vs[0]) + tuple(v.permutedims(p) for v, p in zip(vs[2:], perms[1:])), 300 if n == 32 else 40, (ts[0], ts[1])))
        for dt in (torch.float32, torch.complex128):
            tA = torch.randn(n ** 4, dtype=dt, device="cuda")
            tB = torch.empty_like(tA)
            A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
            for q in ((3, 2, 1, 0), (1, 2, 3, 0)):
                cases.append((f"perm{q} {n}^4 {str(dt)[6:]}", lambda x: x, A.size, (B, A.permutedims(q)), 300 if n == 32 else 40, (tA, tB)))
    for m in (4000, 8192):
        tA = torch.randn(m * m, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        cases.append((f"sym {m}^2 f64", lambda x, y: (x + y) / 2, (m, m), (B, A, A.adjoint()), 20, (tA, tB)))
        cases.append((f"transpose {m}^2 f64", lambda x: x, (m, m), (B, A.permutedims((1, 0))), 20, (tA, tB)))
    for label, f, dims, arrs, reps, keep in cases:
        row, outs = [], []
        for nts in (0, 1):
            S.set_option("nt_store", nts)
            plan = S.make_plan(f, None, None, dims, arrs)
            keep[1].zero_()
            us = time_plan(plan, reps)
            outs.append(keep[1].clone())
            row.append(f"nts={nts}: {us:9.2f} us {plan.algorithmic_bytes / us / 1e3:7.1f} GB/s")
        d = plan.describe()
        print(f"{label:22s} | " + " | ".join(row) + f" | {'same' if torch.equal(outs[0], outs[1]) else 'DIFFERENT'} | {d[d.find('family='):d.find(' ct=')]}")
        sys.stdout.flush()
    # producer -> consumer chains: does a non-temporal producer slow the kernel that reads its output?
    for label, n, dt, q in (("perm chain 32^4 f64", 32, torch.float64, (3, 2, 1, 0)), ("perm chain 64^4 f64", 64, torch.float64, (3, 2, 1, 0)),
                            ("copy chain 2048^2 f32 (as 4-D)", 0, torch.float32, None), ("copy chain 1024^2 f32", -1, torch.float32, None)):
        if n > 0:
            dims = (n,) * 4
        else:
            dims = (2048, 2048) if n == 0 else (1024, 1024)
        numel = 1
        for d in dims:
            numel *= d
        t = [torch.randn(numel, dtype=dt, device="cuda") for _ in range(3)]
        v = [colmajor_view(S, x, dims) for x in t]
        row = []
        for nts in (0, 1):
            S.set_option("nt_store", nts)
            src = (lambda k: v[k].permutedims(q)) if q else (lambda k: v[k])
            p1 = S.make_plan(lambda x: x * 2, None, None, dims, (v[1], src(0)))
            p2 = S.make_plan(lambda x: x + 1, None, None, dims, (v[2], src(1)))

            def both():
                p1.execute(cur())
                p2.execute(cur())
            both()
            torch.cuda.synchronize()
            g = graph_of(torch, both, 100)
            g.replay()
            torch.cuda.synchronize()
            us = min(event_time_ms(torch, g.replay, 3) for _ in range(4)) / 100 * 1e3
            row.append(f"nts={nts}: {us:8.2f} us / pair")
        print(f"{label:32s} | " + " | ".join(row))
    S.set_option("nt_store", -1)


if __name__ == "__main__":
    main()
