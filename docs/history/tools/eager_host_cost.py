#!/usr/bin/env python3
"""Host cost and throughput of back-to-back one-shot launches: a HIP stream (HIP's launch path, every launch ordered) against a
library-owned stream (eager direct dispatch: AQL packets on the library's HSA queues, queue chosen by the data dependencies)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import strided_jl_amd as S


def mk(shape, dt=torch.float64):
    n = 1
    for d in shape:
        n *= d
    t = torch.randn(n, dtype=dt, device="cuda")
    st, s = [], 1
    for d in shape:
        st.append(s); s *= d
    return S.StridedView(t, shape, tuple(st), 0)


def run(name, plans, handle, sync, reps=2000):
    for p in plans:
        p.execute(handle)
    sync()
    best_enq, best_tot = 1e9, 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(reps):
            plans[i % len(plans)].execute(handle)
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        best_enq = min(best_enq, (t1 - t0) / reps * 1e6)
        best_tot = min(best_tot, (t2 - t0) / reps * 1e6)
    print("  %-44s host %6.2f us/call | enqueue + drain %6.2f us/call" % (name, best_enq, best_tot))


def main():
    lib = S.Stream()
    hip = torch.cuda.Stream()
    hsync = lambda: hip.synchronize()  # noqa: E731
    for n in (16, 32):
        A = mk((n,) * 4)
        outs = [mk((n,) * 4) for _ in range(4)]
        perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
        torch.cuda.synchronize()
        chain = [S.make_plan(lambda x: x, None, None, A.size, (outs[0], A.permutedims((3, 2, 1, 0))))]
        indep = [S.make_plan(lambda x: x, None, None, A.size, (o, A.permutedims((3, 2, 1, 0)))) for o in outs]
        step = [chain[0], S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (outs[1],) + tuple(A.permutedims(p) for p in perms))]
        print("%d^4 Float64 (%.2f MiB per array)" % (n, n ** 4 * 8 / 2 ** 20))
        for label, plans in (("permutedims!, same destination (dependent)", chain), ("permutedims!, 4 destinations (independent)", indep),
                             ("bench step: permutedims! + 4-way sum", step)):
            run(label + " | HIP stream", plans, int(hip.cuda_stream), hsync)
            run(label + " | library stream", plans, lib.handle, lib.synchronize)
    print({k: S.get_option("eager_" + k) for k in ("launches", "free", "same", "cross", "fallback")})
    lib.close()


main()
