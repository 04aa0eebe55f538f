#!/usr/bin/env python3
"""Two-sided FLAT form: single elements (flat2_pair = 0) against pairs wherever a row's parity allows (flat2_pair = 2; the default, 1, uses
them for 4-byte elements whose runs are both groups of short leading dims); every row verified."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import strided_jl_amd as S
from reduce_tree_ab import mk, time_plan  # noqa: E402

cases = [((17, 33, 65, 31), (3, 2, 1, 0)), ((5, 300, 300, 7), (3, 2, 1, 0)), ((5, 300, 300, 7), (3, 1, 2, 0)), ((7, 100, 100, 9), (3, 2, 1, 0)), ((3, 500, 500, 3), (3, 2, 1, 0)),
         ((6, 64, 64, 64, 5), (4, 3, 2, 1, 0)), ((10, 200, 200, 10), (3, 2, 1, 0)), ((24, 100, 100, 20), (3, 2, 1, 0)), ((40, 50, 60, 36), (3, 2, 1, 0)),
         ((257, 129, 65), (1, 0, 2)), ((257, 129, 65), (2, 1, 0)), ((2049, 2051), (1, 0)), ((301, 303, 35), (2, 0, 1)), ((12, 5000, 30, 10), (3, 1, 2, 0)), ((31, 29, 10000), (1, 0, 2))]
for dt in (torch.float64, torch.float32, torch.complex64):
    for shape, q in cases:
        A = mk(shape, dt)
        n = len(shape)
        B = mk(tuple(shape[i] for i in q), dt)
        tA = A.parent.reshape(tuple(reversed(shape)))
        ref = tA.permute(*[n - 1 - q[n - 1 - i] for i in range(n)]).contiguous().reshape(-1)
        row, ok, desc = [], True, ""
        for v in (0, 2):
            S.set_option("flat2_pair", v)
            p = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(q)))
            desc = p.describe()
            B.parent.zero_()
            row.append(time_plan(p, 100))
            ok = ok and bool(torch.equal(B.parent, ref))
        S.set_option("flat2_pair", 1)
        if "two-sided" not in desc:
            continue
        b = p.algorithmic_bytes
        print("%-9s %-22s %-16s %6.1f MiB | single %7.2f us %5.2f TB/s | pairs %7.2f us %5.2f TB/s | %s" %
              (str(dt).replace("torch.", ""), shape, q, b / 2 ** 20, row[0], b / row[0] * 1e-6, row[1], b / row[1] * 1e-6, "ok" if ok else "WRONG RESULT"))
        sys.stdout.flush()
