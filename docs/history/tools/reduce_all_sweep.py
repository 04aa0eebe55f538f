#!/usr/bin/env python3
"""Complete reductions (sum of a contiguous array, sum of abs2, dot of two arrays) at 1 MiB ... 1 GiB under different caps on the
number of workgroups (= partials) and with the fold inside the launch.  Usage: python tools/reduce_all_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

lib = S._lib.load()
KEEP = {k: lib.smr_get_option(k.encode()) for k in ("reduce_blocks", "reduce_single")}


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3


for dt in (torch.float64, torch.float32):
    for lg in (17, 19, 20, 21, 22, 24, 27):
        n = 1 << lg
        tA = torch.randn(n, dtype=dt, device="cuda")
        A = colmajor_view(S, tA, (n,))
        out = A.similar(size=(1,))
        ref = float(tA.double().sum())
        row = []
        for blocks, single in ((KEEP["reduce_blocks"], KEEP["reduce_single"]), (64, 0), (128, 0), (256, 0), (512, 0), (1024, 0), (2048, 0), (4096, 0), (64, 1 << 20), (256, 1 << 20)):
            S._lib.check(lib.smr_set_option(b"reduce_blocks", blocks))
            S._lib.check(lib.smr_set_option(b"reduce_single", single))
            plan = S.make_plan(lambda x: x, "+", "zero", (n,), S.promoteshape((n,), out, A))
            us = time_plan(plan, max(3, min(100, int(4e8 / max(plan.algorithmic_bytes, 1)))))
            err = abs(float(out.toarray()[0]) - ref) / max(1.0, abs(ref))
            d = plan.describe()
            row.append("%d%s:%.2f%s" % (blocks, "s" if single else "", us, "" if err < (1e-4 if dt == torch.float32 else 1e-10) else "!"))
        print("sum %-8s %7.1f MiB %s | %s" % (str(dt)[6:], n * tA.element_size() / 2 ** 20, d[d.find("blocks="):d.find(" algbytes")], "  ".join(row)))
        sys.stdout.flush()
for k, v in KEEP.items():
    S._lib.check(lib.smr_set_option(k.encode(), v))
