#!/usr/bin/env python3
"""Ablation of the tiled kernel's phases on the headline workloads (profiling aid; results of
the ablated launches are wrong by construction).  bit0 = no global loads, bit1 = no LDS
exchange/barrier, bit2 = no stores."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402
from tune import time_plan  # noqa: E402

n = 32
tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
tB = torch.empty_like(tA)
A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
work = {
    "perm4321": (lambda x: x, (B, A.permutedims((3, 2, 1, 0)))),
    "bcast4": (lambda a, b, c, d: a + b + c + d, (B,) + tuple(A.permutedims(q) for q in perms)),
    "sym": (lambda x, y: (x + y) / 2, (B.sreshape((1024, 1024)), A.sreshape((1024, 1024)), A.sreshape((1024, 1024)).adjoint())),
}
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] == "big":   # 4096-element tiles on 1024 threads
    S.set_option("max_lds_bytes", 160 * 1024)
    S.set_option("tile_log2", 12)
    for i, v in enumerate((3, 3, 3, 3)):
        S.set_option(f"tile_lg{i}", v)
    work = {"bcast4": work["bcast4"]}
names = {0: "full", 4: "no-store", 2: "no-lds", 1: "no-load", 6: "loads only", 5: "lds only", 3: "stores only", 7: "prologue only"}
for name, (f, arrays) in work.items():
    for ab in (0, 4, 2, 1, 6, 5, 3, 7):
        S.set_option("tiled_ablate", ab)
        plan = S.make_plan(f, None, None, arrays[0].size, arrays)
        us = time_plan(plan, 200)
        print(f"{name:9s} ablate={ab} {names[ab]:14s} {us:8.2f} us")
S.set_option("tiled_ablate", 0)
