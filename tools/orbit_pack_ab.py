#!/usr/bin/env python3
"""Round 6: the ORBIT family with and without shared workgroups for diagonal orbits (option orbit_pack), results checked against NumPy.

HIP events around a hipGraph of 500 launches of ONE kernel (min of 9 replays) -- what bench.py's roofline object reports -- and the
library's own replay (smr_seq: one queue / cut in two).  Sizes: the 4-way permuted sum at 16^4 .. 64^4 (Float64, Float32,
ComplexF64) and the symmetrisation B = (A + A')/2 at 4000^2 / 8192^2."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

dev = torch.device("cuda", 0)
TD = {"f64": torch.float64, "f32": torch.float32, "c64": torch.complex128}


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def timed(plan, reps=500):
    g = graph_of(torch, lambda: plan.execute(cur()), reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(9)) / reps * 1e3


def replay(plan, queues, slices, n=2000):
    st = S.Stream()
    q = S.Sequence().add(plan)
    q.set("queues", queues)
    q.set("slices", slices)
    q.run(50, st.handle)
    q.wait()
    best = 1e30
    for _ in range(7):
        torch.cuda.synchronize()
        t = time.perf_counter()
        q.run(n, st.handle)
        q.wait()
        best = min(best, time.perf_counter() - t)
    del q
    st.close()
    return best / n * 1e6


def sum4(n, dt):
    tA = torch.randn(n ** 4, dtype=TD[dt], device=dev)
    tC = torch.zeros_like(tA)
    A, C = (colmajor_view(S, t, (n,) * 4) for t in (tA, tC))
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]

    def mk():
        return S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in perms))

    a = tA.cpu().numpy().reshape((n,) * 4, order="F")
    want = ((a + a.transpose(perms[1])) + a.transpose(perms[2])) + a.transpose(perms[3])
    return mk, tC, want, 2 * tA.element_size() * n ** 4


def sym(n, dt):
    tA = torch.randn(n * n, dtype=TD[dt], device=dev)
    tB = torch.zeros_like(tA)
    A, B = (colmajor_view(S, t, (n, n)) for t in (tA, tB))

    def mk():
        return S.make_plan(lambda x, y: (x + y) / 2, None, None, A.size, (B, A, A.permutedims((1, 0))))

    a = tA.cpu().numpy().reshape((n, n), order="F")
    return mk, tB, (a + a.T) / 2, 2 * tA.element_size() * n * n


print(torch.cuda.get_device_name(0))
cases = [("sum4 32^4 f64", lambda: sum4(32, "f64")), ("sum4 32^4 f32", lambda: sum4(32, "f32")), ("sum4 32^4 c64", lambda: sum4(32, "c64")),
         ("sum4 16^4 f64", lambda: sum4(16, "f64")), ("sum4 24^4 f64", lambda: sum4(24, "f64")), ("sum4 48^4 f64", lambda: sum4(48, "f64")),
         ("sum4 64^4 f64", lambda: sum4(64, "f64")), ("sym 4000^2 f64", lambda: sym(4000, "f64")), ("sym 8192^2 f64", lambda: sym(8192, "f64"))]
if len(sys.argv) > 1:
    cases = [c for c in cases if any(k in c[0] for k in sys.argv[1:])]
for name, make in cases:
    mk, tout, want, nbytes = make()
    line = "%-16s" % name
    for pack in (0, 1):
        S.set_option("orbit_pack", pack)
        plan = mk()
        tout.zero_()
        plan.execute(cur())
        torch.cuda.synchronize()
        got = tout.cpu().numpy().reshape(want.shape, order="F")
        ok = np.array_equal(got, want)
        us = timed(plan)
        extra = ""
        if "32^4 f64" in name:
            extra = " (replay 1q %.3f, cut in two %.3f)" % (replay(plan, 1, 1), replay(plan, 2, 2))
        d = plan.describe()
        grid = d.split("grid=")[1].split()[0] if "grid=" in d else "?"
        line += " | pack=%d grid=%-5s %8.3f us %6.0f GB/s%s %s" % (pack, grid, us, nbytes / us / 1e3, extra, "ok" if ok else "WRONG")
        del plan
    S.set_option("orbit_pack", 1)
    print(line, flush=True)
