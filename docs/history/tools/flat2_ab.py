#!/usr/bin/env python3
"""Transposing copies whose unit-stride dims are short and not powers of two on BOTH sides: two-sided FLAT form (round 3) against
what ran before (option flat2 = 0), as fallback behind the one-sided form (1, the default) and ahead of it (2).
Usage: python tools/flat2_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

lib = S._lib.load()
KEEP = lib.smr_get_option(b"flat2")


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(5)) / reps * 1e3


cases = [((9, 11, 100000), (1, 0, 2)), ((5, 9, 200000), (1, 0, 2)), ((17, 23, 20000), (1, 0, 2)), ((9, 11, 300, 300), (1, 0, 3, 2)), ((31, 29, 10000), (1, 0, 2)),
         ((12, 10, 14, 9, 11), (4, 3, 2, 1, 0))] if "--batched" in sys.argv else [((17, 33, 65, 31), (3, 2, 1, 0)), ((5, 300, 300, 7), (3, 2, 1, 0))]
for dt in (torch.float64, torch.float32):
    for shape, q in cases:
        N = 1
        for d in shape:
            N *= d
        tA = torch.randn(N, dtype=dt, device="cuda")
        tB = torch.empty_like(tA)
        A = colmajor_view(S, tA, shape)
        dshape = tuple(shape[i] for i in q)
        B = colmajor_view(S, tB, dshape)
        n = len(shape)
        ref = tA.reshape(tuple(reversed(shape))).permute(*[n - 1 - q[n - 1 - i] for i in range(n)]).contiguous().reshape(-1)
        row = []
        # round 4: the batched form (contiguous blocks on both sides) ahead of the forms of round 3 ("flatb" = 0: as in round 3)
        for mode, lead, rb in ((-1, 512, 384), (0, 512, 384), (1, 512, 384), (2, 512, 384)):
            S._lib.check(lib.smr_set_option(b"flatb", 1 if mode < 0 else 0))
            mode = max(mode, 1) if mode < 0 else mode
            S._lib.check(lib.smr_set_option(b"flat2", mode))
            S._lib.check(lib.smr_set_option(b"flat2_lead_bytes", lead))
            S._lib.check(lib.smr_set_option(b"flat2_bytes", rb))
            plan = S.make_plan(lambda x: x, None, None, dshape, (B, A.permutedims(q)))
            tB.zero_()
            us = time_plan(plan, 50)
            ok = torch.equal(tB, ref)
            d = plan.describe()
            lab = "flatb" if "batched" in d else ("flat2" if "two-sided" in d else d[d.find("family=") + 7:d.find(" ct=")])
            row.append("%s/%d/%d %-5s %6.2f%s" % (mode, lead, rb, lab, us, "" if ok else " WRONG"))
        S._lib.check(lib.smr_set_option(b"flatb", 1))
        S._lib.check(lib.smr_set_option(b"flat2", KEEP))
        S._lib.check(lib.smr_set_option(b"flat2_lead_bytes", 512))
        S._lib.check(lib.smr_set_option(b"flat2_bytes", 384))
        print("%-10s %-22s %-16s %5.1f MiB | %s" % (str(dt)[6:], shape, q, 2 * tA.element_size() * N / 2 ** 20, " | ".join(row)))
        sys.stdout.flush()
        del tA, tB
