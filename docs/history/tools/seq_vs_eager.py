#!/usr/bin/env python3
"""The same work -- four chains of 32^4 permutedims! into four destinations (independent of one another), and the bench step -- through a
recorded sequence (smr_seq) and through eager direct dispatch on a library-owned stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import strided_jl_amd as S

n = 32
tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
outs = [torch.empty_like(tA) for _ in range(4)]
st = (1, n, n * n, n ** 3)
V = lambda t: S.StridedView(t, (n,) * 4, st, 0)  # noqa: E731
A = V(tA)
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
pl = [S.make_plan(lambda x: x, None, None, A.size, (V(o), A.permutedims((3, 2, 1, 0)))) for o in outs]
p3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (V(outs[1]),) + tuple(A.permutedims(p) for p in perms))
torch.cuda.synchronize()
lib = S.Stream()
cur = lambda: int(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def seq_time(plans, reps, queues=4):
    q = S.Sequence()
    for p in plans:
        q.add(p)
    q.set("queues", queues)
    q.run(5, cur()); q.wait()
    best = 1e9
    for _ in range(7):
        torch.cuda.synchronize()
        t = time.perf_counter()
        q.run(reps, cur()); q.wait()
        best = min(best, time.perf_counter() - t)
    return best / (reps * len(plans)) * 1e6, q.info().split(" last_replay")[0]


def eager_time(plans, reps):
    for p in plans:
        p.execute(lib.handle)
    lib.synchronize()
    best = 1e9
    for _ in range(7):
        t = time.perf_counter()
        for _ in range(reps):
            for p in plans:
                p.execute(lib.handle)
        lib.synchronize()
        best = min(best, time.perf_counter() - t)
    return best / (reps * len(plans)) * 1e6


for name, plans in (("1 chain (same destination)", pl[:1]), ("2 chains", pl[:2]), ("4 chains", pl), ("bench step", [pl[0], p3])):
    for qn in (1, 2, 4):
        us, info = seq_time(plans, 250, qn)
        print("%-28s seq, %d queue(s): %6.3f us per launch | %s" % (name, qn, us, info))
    print("%-28s eager            : %6.3f us per launch" % (name, eager_time(plans, 250)))
lib.close()
