#!/usr/bin/env python3
"""Partial reductions (family REDUCE_PART) against torch on the same buffers (GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def tm(fn, n=5):
    fn()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, fn, n) for _ in range(3)) * 1e3


TXLOGS = [int(x) for x in os.environ.get('TXLOGS', '5').split(',')]
WGS = [int(x) for x in os.environ.get('WGS', '4096').split(',')]


def main():
    fn = S.fn
    for shape, dims, dt in (((8192, 8192), (0,), torch.float32), ((8192, 8192), (1,), torch.float32),
                            ((4096, 4096), (0,), torch.float64), ((4096, 4096), (1,), torch.float64),
                            ((128, 128, 128, 128), (0, 2), torch.float32), ((128, 128, 128, 128), (1, 3), torch.float32),
                            ((128, 128, 128, 128), (3,), torch.float32), ((64, 64, 64, 64, 8), (0, 1, 2), torch.float64),
                            ((16, 1 << 22), (1,), torch.float32), ((1 << 22, 16), (0,), torch.float32)):
        n = 1
        for d in shape:
            n *= d
        t = torch.randn(n, dtype=dt, device="cuda")
        A = colmajor_view(S, t, shape)
        osz = tuple(1 if i in dims else d for i, d in enumerate(shape))
        on = 1
        for d in osz:
            on *= d
        o = torch.zeros(on, dtype=dt, device="cuda")
        O = colmajor_view(S, o, osz)
        best = None
        for txl in TXLOGS:
            for wgs in WGS:
                S.set_option("reduce_col_txlog", txl)
                S.set_option("reduce_part_wgs", wgs)
                plan = S.make_plan(lambda x: x, "+", None, A.size, S.promoteshape(A.size, O, A))
                us1 = tm(lambda: plan.execute(cur()))
                if len(TXLOGS) * len(WGS) > 1:
                    print(f"    txlog<={txl} wgs={wgs}: {us1:9.1f} us")
                if best is None or us1 < best[0]:
                    best = (us1, plan)
        us, plan = best
        # torch: same memory, row-major view = reversed dims
        tv = t.view(tuple(reversed(shape)))
        tdims = tuple(len(shape) - 1 - d for d in dims)
        ust = tm(lambda: torch.sum(tv, dim=tdims))
        nb = t.element_size() * (n + on)
        d = plan.describe()
        print(f"sum {str(shape):24s} dims={str(dims):10s} {str(dt)[6:]:8s} smr {us:9.1f} us {nb / us / 1e3:7.1f} GB/s | torch {ust:9.1f} us {nb / ust / 1e3:7.1f} GB/s | {d[d.find('nout'):d.find(' algbytes')]}")
        sys.stdout.flush()
        del t, o
    # __mul!-shaped 3-operand reduction: C[i,j] += sum_k A[i,k] * B[k,j]  (src/linalg.jl:130-162)
    for m in (512, 1024):
        ta, tb, tc = (torch.randn(m * m, dtype=torch.float32, device="cuda") for _ in range(3))
        A, B, C = (colmajor_view(S, x, (m, m)) for x in (ta, tb, tc))
        A3 = S.StridedView(A.parent, (m, m, m), (1, 0, m), 0)   # A[i,k]: dims (i, j, k)
        B3 = S.StridedView(B.parent, (m, m, m), (0, m, 1), 0)   # B[k,j]
        C3 = S.StridedView(C.parent, (m, m, m), (1, m, 0), 0)
        plan = S.make_plan(lambda a, b: a * b, "+", "zero", (m, m, m), (C3, A3, B3))
        us = tm(lambda: plan.execute(cur()), 2)
        ust = tm(lambda: torch.mm(ta.view(m, m), tb.view(m, m)), 5)
        print(f"mul {m}^3 f32 smr {us:9.1f} us {2 * m ** 3 / us / 1e6:7.2f} TFLOP/s | torch.mm {ust:8.1f} us | {plan.describe()}")


if __name__ == "__main__":
    main()
