#!/usr/bin/env python3
"""Split reductions in one launch (the workgroup arriving last folds the partials; option reduce_single=1, round 3)
against the two-launch form (reduce_single=0): time per reduction, results checked against torch in float64.
Usage: python tools/reduce_single_ab.py"""
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
KEEP = lib.smr_get_option(b"reduce_single")


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def time_plan(plan, reps):
    plan.execute(cur())
    torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(5)) / reps * 1e3


def run(label, dims, dt, rd):
    n = int(np.prod(dims))
    tA = torch.randn(n, dtype=dt, device="cuda")
    A = colmajor_view(S, tA, dims)
    odims = tuple(1 if d in rd else m for d, m in enumerate(dims))
    out = A.similar(size=odims)
    nd = len(dims)
    ref = tA.reshape(tuple(reversed(dims))).to(torch.float64 if not dt.is_complex else torch.complex128).sum(dim=[nd - 1 - d for d in rd]).reshape(-1)
    row = []
    for single in (1 << 20, 0):
        S._lib.check(lib.smr_set_option(b"reduce_single", single))
        plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
        reps = max(3, min(100, int(4e8 / max(plan.algorithmic_bytes, 1))))
        us = time_plan(plan, reps)
        got = torch.as_tensor(out.toarray()).reshape(tuple(odims)).permute(*reversed(range(nd))).reshape(-1).to(ref.dtype).cuda() if False else None
        res = torch.from_numpy(np.ascontiguousarray(out.toarray().transpose(tuple(reversed(range(nd)))))).reshape(-1).to(ref.dtype).cuda()
        err = float((res - ref).abs().max() / max(1e-30, float(ref.abs().max())))
        tol = 1e-5 if dt in (torch.float32, torch.complex64) else 1e-12
        row.append("%7.2f us %5.0f GB/s%s" % (us, plan.algorithmic_bytes / us / 1e3, "" if err < tol else " WRONG(%.2e)" % err))
        d = plan.describe()
    S._lib.check(lib.smr_set_option(b"reduce_single", KEEP))
    print("%-40s %6.1f MiB | one launch %s | two launches %s | %s" % (label, plan.algorithmic_bytes / 2 ** 20, row[0], row[1], d[d.find("nout="):d.find(" algbytes")]))
    sys.stdout.flush()


for dims, dt in (((100, 90, 80, 7), torch.float32), ((512, 384, 64), torch.float32), ((100, 90, 80, 7), torch.float64), ((64, 48, 40, 30), torch.complex128)):
    for k in range(1, len(dims)):
        for rd in itertools.combinations(range(len(dims)), k):
            run("sum %s %s dims=%s" % (str(dt)[6:], dims, rd), dims, dt, rd)
# complete reductions
for n, dt in ((32 ** 4, torch.float64), (1 << 22, torch.float32), (1 << 24, torch.float64), (1 << 27, torch.float64)):
    tA = torch.randn(n, dtype=dt, device="cuda")
    A = colmajor_view(S, tA, (n,))
    out = A.similar(size=(1,))
    row = []
    for single in (1 << 20, 0):
        S._lib.check(lib.smr_set_option(b"reduce_single", single))
        plan = S.make_plan(lambda x: x, "+", "zero", (n,), S.promoteshape((n,), out, A))
        reps = max(3, min(100, int(4e8 / max(plan.algorithmic_bytes, 1))))
        us = time_plan(plan, reps)
        err = abs(float(out.toarray()[0]) - float(tA.double().sum())) / max(1.0, abs(float(tA.double().sum())))
        row.append("%7.2f us %5.0f GB/s%s" % (us, plan.algorithmic_bytes / us / 1e3, "" if err < (1e-4 if dt == torch.float32 else 1e-10) else " WRONG(%.2e)" % err))
    S._lib.check(lib.smr_set_option(b"reduce_single", KEEP))
    print("%-40s %6.1f MiB | one launch %s | two launches %s" % ("sum %s (%d,)" % (str(dt)[6:], n), plan.algorithmic_bytes / 2 ** 20, row[0], row[1]))
