#!/usr/bin/env python3
"""what do the planner's knobs give for the partial sums of (100, 90, 80, 7) Float64 and friends?
Every dim subset under the planner's choice and under overrides; best five per case.  Usage: python tools/item4_sweep.py"""
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
NAMES = ("reduce_part_wgs", "reduce_col_txlog", "reduce_part_kind", "reduce_single", "reduce_col_narrow", "reduce_row_floor", "reduce_row_dense")
DEFAULTS = {k: lib.smr_get_option(k.encode()) for k in NAMES}


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps=40):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3


def setopts(**kw):
    for k, v in dict(DEFAULTS, **kw).items():
        S._lib.check(lib.smr_set_option(k.encode(), v))


shapes = [((100, 90, 80, 7), torch.float64)]
if "--more" in sys.argv:
    shapes += [((100, 90, 80, 7), torch.float32), ((512, 384, 64), torch.float32), ((20, 30, 40, 50), torch.float64)]
for dims, dt in shapes:
    n = int(np.prod(dims))
    A = colmajor_view(S, torch.randn(n, dtype=dt, device="cuda"), dims)
    for k in range(1, len(dims)):
        for rd in itertools.combinations(range(len(dims)), k):
            out = A.similar(size=tuple(1 if d in rd else m for d, m in enumerate(dims)))
            res, seen = [], set()
            setopts()
            plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
            d = plan.describe()
            base = (time_plan(plan), "defaults: %s" % d[d.find("dims="):d.find(" algbytes")])
            for kind in (-1, 1, 2):
                for wgs in (1024, 128, 256, 512, 2048):
                    for tx in (5, 2, 3, 4, 6):
                        for narrow in (1, 0):
                            for single in (DEFAULTS["reduce_single"], 1 << 20):
                                setopts(reduce_part_kind=kind, reduce_part_wgs=wgs, reduce_col_txlog=tx, reduce_col_narrow=narrow, reduce_single=single)
                                plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
                                d = plan.describe()
                                key = d[d.find("dims="):d.find(" algbytes")] + (":single" if single > 4 and "split=1 " not in d + " " else "")
                                if key in seen:
                                    continue
                                seen.add(key)
                                res.append((time_plan(plan), "kind=%d wgs=%d tx=%d narrow=%d single=%d: %s" % (kind, wgs, tx, narrow, 1 if single > 4 else 0, d[d.find("dims="):d.find(" algbytes")])))
            res.sort()
            print("%s %s dims=%s: planner %.2f us (%.0f GB/s) [%s]" % (str(dt)[6:], dims, rd, base[0], plan.algorithmic_bytes / base[0] / 1e3, base[1]))
            for us, lab in res[:4]:
                print("      %7.2f us  %5.0f GB/s  %s" % (us, plan.algorithmic_bytes / us / 1e3, lab))
            sys.stdout.flush()
setopts()
