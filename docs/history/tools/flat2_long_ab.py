#!/usr/bin/env python3
"""Ragged transposes with LONG unit-stride dims: TILED (power-of-two 32-wide tiles, flat2_long = 0) against the two-sided FLAT form with
evenly cut leads (flat2_long = 80: the default rule; profiles/r04_flat2_long_ab.txt also holds the run with the form applied everywhere)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import strided_jl_amd as S
from reduce_tree_ab import mk, time_plan  # noqa: E402

for dt in (torch.float64, torch.float32):
    for shape in ((100, 90, 80), (257, 129, 65), (17, 33, 65, 31), (48, 36, 24, 30), (1000, 3, 700), (200, 300, 70), (130, 70, 50, 9), (1400, 1500), (999, 1001), (4000, 4100),
                  (1024, 1024), (4096, 4096), (96, 64, 80), (128, 128, 64), (32, 32, 32, 32), (64, 64, 64, 64), (8192, 8192), (96, 96, 96, 96),
                  (2049, 2051), (1001, 1100), (513, 700, 9), (4001, 4003), (301, 303, 35)):
        A = mk(shape, dt)
        n = len(shape)
        for q in itertools.permutations(range(n)):
            if q[0] == 0 or (n == 4 and q not in ((3, 2, 1, 0), (3, 2, 0, 1), (1, 0, 2, 3), (2, 3, 0, 1))):
                continue
            B = mk(tuple(shape[i] for i in q), dt)
            row, descs = [], []
            for v in (0, 80):
                S.set_option("flat2_long", v)
                p = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(q)))
                B.parent.zero_()
                row.append(time_plan(p, 100))
                tA = A.parent.reshape(tuple(reversed(shape)))          # torch sees the column-major array with reversed index order
                ref = tA.permute(*[n - 1 - q[n - 1 - i] for i in range(n)]).contiguous().reshape(-1)
                if not torch.equal(B.parent, ref):
                    print("WRONG RESULT:", shape, q, dt, v, p.describe())
                d = p.describe()
                descs.append(d[d.find("family=") + 7:d.find(" ct=")] + (":2s" if "two-sided" in d else ""))
            S.set_option("flat2_long", 80)
            if descs[0] == descs[1]:
                continue
            b = p.algorithmic_bytes
            print("%-8s %-20s %-14s %6.1f MiB | %-8s %7.2f us %5.2f TB/s | %-8s %7.2f us %5.2f TB/s" %
                  (str(dt).replace("torch.", ""), shape, q, b / 2 ** 20, descs[0], row[0], b / row[0] * 1e-6, descs[1], row[1], b / row[1] * 1e-6))
            sys.stdout.flush()
