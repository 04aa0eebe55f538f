#!/usr/bin/env python3
"""A/B of the ORBIT family (inputs = permuted views of one buffer, option `orbit`) against the classic
tiled kernel, plus the non-temporal-store switch (`nt_store`), on the README workloads (GPU box only).
Every timed plan is first checked bit-for-bit against torch."""
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
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3  # us


def main():
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    sizes = [int(x) for x in os.environ.get("NS", "32,64,128").split(",")]
    cfgs = [dict(orbit=0, nt_store=0), dict(orbit=0, nt_store=1), dict(orbit=1, nt_store=0), dict(orbit=1, nt_store=1)]
    for dt in (torch.float64, torch.float32):
        for n in sizes:
            if dt == torch.float32 and n > 64:
                continue
            reps = 200 if n <= 32 else (40 if n <= 64 else 4)
            tA = torch.randn(n ** 4, dtype=dt, device="cuda")
            tB = torch.empty_like(tA)
            A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
            a4 = tA.reshape((n,) * 4)
            cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
            ref = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)
            for cfg in cfgs:
                for k, v in cfg.items():
                    S.set_option(k, v)
                plan = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
                tB.zero_()
                plan.execute(cur())
                torch.cuda.synchronize()
                ok = torch.equal(tB, ref)
                us = time_plan(plan, reps)
                d = plan.describe()
                print(f"bcast4 {n}^4 {str(dt)[6:]:8s} orbit={cfg['orbit']} nts={cfg['nt_store']} {us:10.2f} us {2 * tA.element_size() * n ** 4 / us / 1e3:8.1f} GB/s "
                      f"{'ok' if ok else 'WRONG'} | {d[d.find('family='):d.find(' ct=')]} {d[d.find('tile='):d.find(' algbytes')]}")
                sys.stdout.flush()
            del tA, tB, ref
    for m in (4000, 8192):
        for dt in (torch.float64, torch.complex64):
            tA = torch.randn(m * m, dtype=dt, device="cuda")
            tB = torch.empty_like(tA)
            A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
            am = tA.reshape(m, m)  # row-major view of column-major data = the transpose
            for name, views, reff in (("sym", (A, A.permutedims((1, 0))), lambda: ((am + am.t()) / 2)),
                                      ("herm", (A, A.adjoint()), lambda: ((am + am.t().conj()) / 2))):
                if name == "herm" and dt != torch.complex64:
                    continue
                ref = reff().contiguous().reshape(-1)
                for cfg in cfgs:
                    for k, v in cfg.items():
                        S.set_option(k, v)
                    plan = S.make_plan(lambda x, y: (x + y) / 2, None, None, (m, m), (B,) + views)
                    tB.zero_()
                    plan.execute(cur())
                    torch.cuda.synchronize()
                    ok = torch.equal(tB, ref)
                    us = time_plan(plan, 20)
                    d = plan.describe()
                    print(f"{name:5s} {m}^2 {str(dt)[6:]:9s} orbit={cfg['orbit']} nts={cfg['nt_store']} {us:10.2f} us {2 * tA.element_size() * m * m / us / 1e3:8.1f} GB/s "
                          f"{'ok' if ok else 'WRONG'} | {d[d.find('family='):d.find(' ct=')]} {d[d.find('tile='):d.find(' algbytes')]}")
                    sys.stdout.flush()
            del tA, tB
    # permutedims! with / without non-temporal stores
    for n in (32, 64, 128):
        tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        for nts in (0, 1):
            S.set_option("nt_store", nts)
            plan = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
            us = time_plan(plan, 200 if n <= 32 else (40 if n <= 64 else 4))
            ok = torch.equal(tB, tA.reshape((n,) * 4).permute(3, 2, 1, 0).contiguous().reshape(-1))
            print(f"perm4321 {n}^4 f64 nts={nts} {us:10.2f} us {16 * n ** 4 / us / 1e3:8.1f} GB/s {'ok' if ok else 'WRONG'}")
        del tA, tB
    S.set_option("nt_store", -1)
    S.set_option("orbit", 1)


if __name__ == "__main__":
    main()
