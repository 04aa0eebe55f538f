#!/usr/bin/env python3
"""Round 5: what the fences of a replayed packet cost, and which of them the bench step needs.

The bench step (32^4 Float64: permutedims!(B, A, (4,3,2,1)) and C .= sum of 4 permuted views of A) is replayed by smr_seq under
every combination of
    queues        1 (in order) | 2 (one per dependency component) | 3 (the sum cut in two block ranges) | 4 (both cut in two)
    acquire       agent on every packet (round 4) | by need (none here: nobody in the sequence writes A)
    release       agent (default) | none (EXPERIMENT ONLY: write-after-write across XCDs is not ordered without it)
for long replays (K = 1000: steady state) and the driver's K = 20, wall clock around smr_seq_run + smr_seq_wait, best of 7;
outputs are checked bit for bit against torch after every configuration.

Usage: python tools/fence_ab.py [--n 32]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--long", type=int, default=1000)
args = ap.parse_args()
n = args.n
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1234)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev, generator=g)
tB = torch.empty_like(tA)
tC = torch.empty_like(tA)
A, B, Cc = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
plan2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
plan3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))
a4 = tA.reshape((n,) * 4)
ref2 = a4.permute(3, 2, 1, 0).contiguous().reshape(-1)
cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
ref3 = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


LIBSTREAM = S.Stream()


def measure(q, K):
    q.run(max(2, K // 10), LIBSTREAM.handle); q.wait()
    best = 1e30
    for _ in range(7):
        torch.cuda.synchronize()
        t = time.perf_counter()
        q.run(K, LIBSTREAM.handle); q.wait()
        best = min(best, time.perf_counter() - t)
    return best / K * 1e6


print("%s  n = %d  (algorithmic bytes per step %d)" % (torch.cuda.get_device_name(0), n, 2 * 2 * 8 * n ** 4))
print("%-38s %-9s %-8s | %10s %10s | %s" % ("queues", "acquire", "release", "us/step K=%d" % args.long, "K=20", "bit-exact"))
layouts = [("1 (in order)", dict(queues=1)),
           ("2 (per component)", dict(queues=2)),
           ("3 (perm | sum/2 | sum/2)", {"queues": 3, "slices:1": 2}),
           ("3 (perm/2 | perm/2 | sum)", {"queues": 3, "slices:0": 2}),
           ("4 (perm/2 x2 | sum/2 x2)", dict(queues=4, slices=2)),
           ("4 (perm | sum/3 x3)", {"queues": 4, "slices:1": 3})]
for lname, lay in layouts:
    for acq, aname in ((1, "agent"), (-1, "by-need")):
        for rel, rname in ((1, "agent"), (0, "NONE(exp)")):
            q = S.Sequence().add(plan2).add(plan3)
            for k, v in lay.items():
                q.set(k, v)
            q.set("acquire", acq)
            q.set("release", rel)
            tB.zero_(); tC.zero_()
            torch.cuda.synchronize()
            usl = measure(q, args.long)
            us20 = measure(q, 20)
            torch.cuda.synchronize()
            ok = torch.equal(tB, ref2) and torch.equal(tC, ref3)
            info = q.info()
            print("%-38s %-9s %-8s | %10.3f %10.3f | %s | %s" % (lname, aname, rname, usl, us20, "yes" if ok else "NO",
                                                            " ".join(w for w in info.split() if w.split("=")[0] in ("queues", "sliced", "acquire"))))
            del q
