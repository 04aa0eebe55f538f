#!/usr/bin/env python3
"""Round 5: with the kernel boundaries gone (self-released launches, 4 queues, device 100 % busy) the replayed bench step is bound
by the two kernels sharing the chip -- do other plan shapes of the two kernels do better in THAT regime than the ones tuned for
single launches?  Replays the step (K = 1000, best of 7) with planner options forced; bit-exact check each time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

n = 32
dev = torch.device("cuda", 0)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
tB, tC = torch.empty_like(tA), torch.empty_like(tA)
A, B, Cc = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
a4 = tA.reshape((n,) * 4)
ref2 = a4.permute(3, 2, 1, 0).contiguous().reshape(-1)
cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
ref3 = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)
st = S.Stream()


def measure(q, K=1000):
    q.run(100, st.handle); q.wait()
    best = 1e30
    for _ in range(7):
        torch.cuda.synchronize()
        t = time.perf_counter()
        q.run(K, st.handle); q.wait()
        best = min(best, time.perf_counter() - t)
    return best / K * 1e6


def run(opts2, opts3, seqopts=None):
    saved = {}
    for k, v in opts2.items():
        saved[k] = S.get_option(k); S.set_option(k, v)
    p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    for k, v in saved.items():
        S.set_option(k, v)
    saved = {}
    for k, v in opts3.items():
        saved[k] = S.get_option(k); S.set_option(k, v)
    p3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))
    # options that act at launch / record time stay set while the sequence is built
    q = S.Sequence().add(p2).add(p3)
    for k, v in (seqopts or {}).items():
        q.set(k, v)
    tB.zero_(); tC.zero_()
    torch.cuda.synchronize()
    try:
        us = measure(q)
        torch.cuda.synchronize()
        ok = torch.equal(tB, ref2) and torch.equal(tC, ref3)
        info = q.info()
    except Exception as e:  # noqa: BLE001
        us, ok, info = float("nan"), False, str(e)[:100]
    for k, v in saved.items():
        S.set_option(k, v)
    print("perm %-28s sum %-34s %-22s | %7.3f us/step %s | %s | %s" % (opts2 or "default", opts3 or "default", seqopts or "", us, "ok" if ok else "WRONG",
                                                                       p2.describe().split("tile=")[1][:30], p3.describe().split("family=")[1][:60]), flush=True)
    del q


run({}, {})
run({}, {"orbit_lg": 3})
run({}, {"orbit_lg": 3}, {"slices:1": 1, "queues": 3})
run({}, {"orbit": 0})
run({}, {"orbit": 0, "tile_log2": 12})
run({"tile_log2": 12}, {})
run({"tile_log2": 10}, {})
run({"tiled_persist": 0}, {})
run({}, {"orbit_pipe": 1})
run({}, {"orbit_lds_min": 65536})
run({}, {"orbit_lds_min": 32768})
run({}, {}, {"queues": 3, "slices:1": 2})
run({}, {}, {"queues": 4, "slices:1": 3})
