#!/usr/bin/env python3
"""Planner sweep for mid-sized partial reductions: every dim subset of two shapes, timed under the
planner's choice and under overrides of reduce_part_wgs / reduce_col_txlog / reduce_part_kind.
Usage: python tools/reduce_sweep.py [--quick]"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

lib = S._lib.load()
FULL = "--full" in sys.argv
DEFAULTS = {"reduce_part_wgs": lib.smr_get_option(b"reduce_part_wgs"), "reduce_col_txlog": lib.smr_get_option(b"reduce_col_txlog"), "reduce_part_kind": -1,
            "reduce_single": lib.smr_get_option(b"reduce_single"), "reduce_col_narrow": lib.smr_get_option(b"reduce_col_narrow"), "reduce_row_floor": -1, "reduce_row_dense": 1}


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3


def setopts(**kw):
    for k, v in dict(DEFAULTS, **kw).items():
        S._lib.check(lib.smr_set_option(k.encode(), v))


for dims, dt in (((100, 90, 80, 7), torch.float32), ((512, 384, 64), torch.float32), ((100, 90, 80, 7), torch.float64), ((256, 256, 256), torch.float32),
                 ((64, 64, 64), torch.float64), ((1000, 1000), torch.float64), ((4096, 4096), torch.float32),
                 ((3, 1920, 1080), torch.float32), ((3, 1920, 1080), torch.float64), ((7, 1000000), torch.float64), ((48, 300, 300), torch.complex64),
                 ((64, 100000), torch.float32), ((20, 30, 40, 50), torch.float64), ((12, 400000), torch.float32), ((32, 200000), torch.float32),
                 ((200, 50000), torch.float32), ((1000, 20000), torch.float32), ((7, 1000, 1000), torch.float64), ((4, 512, 512, 8), torch.float32)):
    n = int(np.prod(dims))
    A = colmajor_view(S, torch.randn(n, dtype=dt, device="cuda"), dims)
    for k in range(1, len(dims)):
        for rd in itertools.combinations(range(len(dims)), k):
            out = A.similar(size=tuple(1 if d in rd else m for d, m in enumerate(dims)))
            res = []
            seen = set()
            setopts()
            plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
            d = plan.describe()
            res.append((time_plan(plan, 30), "defaults: %s" % d[d.find("form="):d.find(" algbytes")]))
            setopts(reduce_row_dense=0)
            plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
            d2 = plan.describe()
            if d2 != d:
                res.append((time_plan(plan, 30), "row_dense=0: %s" % d2[d2.find("form="):d2.find(" algbytes")]))
            for fl in (0, 2, 4):
                setopts(reduce_row_floor=fl)
                plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
                d2 = plan.describe()
                if "form=row" in d2:
                    res.append((time_plan(plan, 30), "row_floor=%d: %s" % (fl, d2[d2.find("form="):d2.find(" algbytes")])))
            for kind in (-1, 1, 2):
                for wgs in (1024, 256, 512, 2048, 4096):
                    for tx in (5, 3, 4, 6, 7, 8):
                        setopts(reduce_part_kind=kind, reduce_part_wgs=wgs, reduce_col_txlog=tx, reduce_col_narrow=0)
                        plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
                        d = plan.describe()
                        key = d[d.find("nout="):d.find(" algbytes")] + d[d.find("dims="):d.find(" nout")]
                        if "form=col" not in d and tx != 5:
                            continue
                        if "split=1 " in d + " " and wgs != 4096:
                            continue
                        if key + str(tx) in seen:
                            continue
                        seen.add(key + str(tx))
                        for single in ((0, 1 << 20) if "split=1 " not in d + " " else (DEFAULTS["reduce_single"],)):
                            S._lib.check(lib.smr_set_option(b"reduce_single", single))
                            us = time_plan(plan, 30)
                            res.append((us, "kind=%d wgs=%d tx=%d single=%d: %s" % (kind, wgs, tx, 1 if single else 0, d[d.find("form="):d.find(" algbytes")])))
            base = res[0]
            res.sort()
            print("%s %s dims=%s: planner %.2f us (%.0f GB/s) [%s]" % (str(dt)[6:], dims, rd, base[0], plan.algorithmic_bytes / base[0] / 1e3, base[1]))
            for us, lab in (res if FULL else res[:5]):
                print("      %7.2f us  %s" % (us, lab))
            sys.stdout.flush()
setopts()
