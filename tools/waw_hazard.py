#!/usr/bin/env python3
"""Round 5: is the release fence of a replayed packet needed -- and what replaces it?

Two plans write the SAME destination B through DIFFERENT tilings (B <- permutedims(A1, (4,3,2,1)) and B <- permutedims(A2, (2,1,4,3)),
32^4 ... 64^4 Float64): a given 128-byte line of B is written by a workgroup of one XCD in the first launch and by a workgroup of
(usually) another XCD in the second.  The sequence [P1, P2] is replayed; in-order execution leaves B = permutedims(A2, (2,1,4,3)).
  * plain stores, release fence dropped (experiment: "release" = 0, option seq_self_release = 0): the first launch's lines may still
    sit dirty in ITS XCD's L2 when the second launch has written them through another L2; whichever is written back last wins;
  * default (round 5): both launches are self-released (agent-scope write-through stores, acknowledged before a wave ends) and their
    packets carry no release fence;
  * r04 form: plain / non-temporal stores and an agent-scope release on every packet.
Prints the number of wrong elements per form (0 expected for the last two)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

dev = torch.device("cuda", 0)


def run(n, form, reps=50):
    g = torch.Generator(device=dev)
    g.manual_seed(n)
    t1 = torch.randn(n ** 4, dtype=torch.float64, device=dev, generator=g)
    t2 = torch.randn(n ** 4, dtype=torch.float64, device=dev, generator=g)
    tb = torch.zeros_like(t1)
    A1, A2, B = (colmajor_view(S, t, (n,) * 4) for t in (t1, t2, tb))
    S.set_option("seq_self_release", 0 if form != "self-released" else 1)
    S.set_option("nt_store", 0 if form == "plain, no release" else -1)
    p1 = S.make_plan(lambda x: x, None, None, B.size, (B, A1.permutedims((3, 2, 1, 0))))
    p2 = S.make_plan(lambda x: x, None, None, B.size, (B, A2.permutedims((1, 0, 3, 2))))
    q = S.Sequence().add(p1).add(p2)
    if form == "plain, no release":
        q.set("release", 0)
    st = S.Stream()
    wrong = 0
    a4 = t2.reshape((n,) * 4)
    p = (1, 0, 3, 2)
    want = a4.permute(*[3 - p[3 - i] for i in range(4)]).contiguous().reshape(-1)
    for r in range(reps):
        q.run(3, st.handle); q.wait()
        torch.cuda.synchronize()
        wrong += int((tb != want).sum().item())
        tb.zero_()
        torch.cuda.synchronize()
    info = q.info()
    del q
    st.close()
    S.set_option("seq_self_release", 1)
    S.set_option("nt_store", -1)
    return wrong, info


if __name__ == "__main__":
    for n in (32, 48, 64):
        for form in ("plain, no release", "self-released", "agent release"):
            wrong, info = run(n, form)
            keys = ("queues", "acquire", "release", "self_released")
            print("%2d^4  %-20s wrong elements over 50 runs: %9d   | %s" % (n, form, wrong, " ".join(w for w in info.split() if w.split("=")[0] in keys)), flush=True)
