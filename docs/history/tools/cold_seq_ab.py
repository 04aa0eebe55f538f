#!/usr/bin/env python3
"""Round 5: the HBM-cold form of the headline kernels (40 rotating (A, B) pairs = 640 MiB) as a 2-queue sequence: acquire by need
(none here) against agent-scope acquire on every packet (round 4), release agent / self-released forced on."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

dev = torch.device("cuda", 0)
n, npair = 32, 40
poolA = torch.randn(npair, n ** 4, dtype=torch.float64, device=dev)
poolB = torch.empty_like(poolA)
A, B = colmajor_view(S, poolA[0], (n,) * 4), colmajor_view(S, poolB[0], (n,) * 4)
esz = poolA.element_size() * n ** 4
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
st = S.Stream()
for name, f, srcs in (("permutedims", lambda x: x, (A.permutedims((3, 2, 1, 0)),)),
                      ("4-way sum", lambda a, b, c, d: a + b + c + d, tuple(A.permutedims(q) for q in perms))):
    p = S.make_plan(f, None, None, A.size, (B,) + srcs)
    for label, opts, total in (("acquire by need, release agent (default for this footprint)", {}, None),
                               ("acquire agent on every packet (round 4)", {"acquire": 1}, None),
                               ("self-released forced (write-through, no release), acquire by need", {}, 1 << 40),
                               ("... and acquire agent", {"acquire": 1}, 1 << 40)):
        if total:
            S.set_option("self_release_max_total", total)
        q = S.Sequence()
        for i in range(npair):
            q.add(p, bases=[poolB.data_ptr() + i * esz] + [poolA.data_ptr() + i * esz] * len(srcs))
        q.set("queues", 2)
        for k, v in opts.items():
            q.set(k, v)
        q.run(2, st.handle); q.wait()
        best = 1e30
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q.run(8, st.handle); q.wait()
            best = min(best, time.perf_counter() - t0)
        info = q.info()
        S.set_option("self_release_max_total", 128 << 20)
        print("%-12s %-70s %7.3f us per launch | %s" % (name, label, best / (8 * npair) * 1e6, " ".join(w for w in info.split() if w.split("=")[0] in ("queues", "acquire", "self_released"))), flush=True)
        del q
