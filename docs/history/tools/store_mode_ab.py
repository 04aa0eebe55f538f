#!/usr/bin/env python3
"""Round 5 experiment: store policy of the two headline kernels x release fence of the replayed packets.

nt_store = -1 automatic (TILED non-temporal, ORBIT plain at 32^4) / 0 plain / 1 non-temporal / 2 agent-scope WRITE-THROUGH
(`global_store ... sc1`, the wave waits for the acknowledgements before it ends: smr_device.h, store policy 2), with the sequences'
own choice of policy switched off (option seq_self_release = 0) so that the forced policy is what runs.  For each, release in
{agent, none} and the bench step (32^4 Float64) replayed on 2 / 3 / 4 queues: us per step (K = 1000 and K = 20, best of 7), bit-exact
check after every configuration.  Last block: the library's default (self-released launches, no release fence on their packets).
(The first version of this experiment used a second build of the library in which the "non-temporal" stores were compiled as sc1
stores; the numbers of that run are kept in profiles/r05_store_mode_ab_sc1_build.txt.)
The question: if a kernel's stores are written through to the memory side (and acknowledged before the wave ends), the packet's
release fence -- a write-back of all eight L2s by the packet processor -- has nothing left to do; is the step faster without it?
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

n = 32
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1234)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev, generator=g)
tB = torch.empty_like(tA)
tC = torch.empty_like(tA)
A, B, Cc = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
a4 = tA.reshape((n,) * 4)
ref2 = a4.permute(3, 2, 1, 0).contiguous().reshape(-1)
cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
ref3 = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)
st = S.Stream()


def measure(q, K):
    q.run(max(2, K // 10), st.handle); q.wait()
    best = 1e30
    for _ in range(7):
        torch.cuda.synchronize()
        t = time.perf_counter()
        q.run(K, st.handle); q.wait()
        best = min(best, time.perf_counter() - t)
    return best / K * 1e6


print("%-8s %-28s %-9s | %9s %9s | %s" % ("nt_store", "queues", "release", "us/step", "K=20", "bit-exact / plans"))
S.set_option("seq_self_release", 0)
for nt in (-1, 0, 1, 2):
    S.set_option("nt_store", nt)
    plan2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    plan3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))
    for lname, lay in (("2 (per component)", {"queues": 2, "slices": 1}), ("3 (perm | sum/2 x2)", {}), ("4 (perm/2 x2 | sum/2 x2)", {"queues": 4, "slices": 2})):
        for rel in (1, 0):
            q = S.Sequence().add(plan2).add(plan3)
            for k, v in lay.items():
                q.set(k, v)
            q.set("release", rel)
            tB.zero_(); tC.zero_()
            torch.cuda.synchronize()
            us = measure(q, 1000)
            us20 = measure(q, 20)
            torch.cuda.synchronize()
            ok = torch.equal(tB, ref2) and torch.equal(tC, ref3)
            print("%-8d %-28s %-9s | %9.3f %9.3f | %s" % (nt, lname, "agent" if rel else "NONE", us, us20, "yes" if ok else "NO"), flush=True)
            del q
    del plan2, plan3
S.set_option("nt_store", -1)
S.set_option("seq_self_release", 1)
print("library default: launches recorded for a sequence are self-released (write-through stores), their packets carry no release fence")
plan2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
plan3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))
for lname, lay in (("2 (per component)", {"queues": 2, "slices": 1}), ("3 (perm | sum/2 x2)", {"queues": 3, "slices:1": 2}), ("4 (perm/2 x2 | sum/2 x2)", {"queues": 4, "slices": 2}),
                   ("automatic", {}), ("automatic, release fences kept", {"release_self": 1})):
    q = S.Sequence().add(plan2).add(plan3)
    for k, v in lay.items():
        q.set(k, v)
    tB.zero_(); tC.zero_()
    torch.cuda.synchronize()
    us = measure(q, 1000)
    us20 = measure(q, 20)
    torch.cuda.synchronize()
    ok = torch.equal(tB, ref2) and torch.equal(tC, ref3)
    print("%-8s %-28s %-9s | %9.3f %9.3f | %s | %s" % ("default", lname, "by-need", us, us20, "yes" if ok else "NO", q.info()[:250]), flush=True)
    del q
