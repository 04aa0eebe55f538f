#!/usr/bin/env python3
"""The bench step (permutedims! then the 4-way sum, alternating in one graph) against its two kernels timed alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def t(fn, reps=500):
    fn()
    torch.cuda.synchronize()
    g = graph_of(torch, fn, reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(9)) / reps * 1e3


n = 32
tA = torch.randn(n ** 4, dtype=torch.float64, device="cuda")
tB = torch.empty_like(tA)
tC = torch.empty_like(tA)
A, B, C = (colmajor_view(S, x, (n,) * 4) for x in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
for nts in (-1, 0):
    S.set_option("nt_store", nts)
    p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    p3c = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in perms))
    p3b = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(p) for p in perms))
    a2, a3 = t(lambda: p2.execute(cur())), t(lambda: p3c.execute(cur()))

    def step_c():
        p2.execute(cur())
        p3c.execute(cur())

    def step_b():
        p2.execute(cur())
        p3b.execute(cur())

    def pair22():
        p2.execute(cur())
        p2.execute(cur())

    print(f"nt_store={nts}: perm alone {a2:.2f} us, sum4 alone {a3:.2f} us, sum {a2 + a3:.2f} | step (sum4 -> C) {t(step_c):.2f} | step (sum4 -> B) {t(step_b):.2f} | perm,perm {t(pair22):.2f}")
S.set_option("nt_store", -1)
# which pairs pay for alternating?
tD = torch.empty_like(tA)
D = colmajor_view(S, tD, (n,) * 4)
pc = S.make_plan(lambda x: x, None, None, A.size, (D, A))                                   # STREAM copy
p2b = S.make_plan(lambda x: x, None, None, A.size, (C, A.permutedims((1, 2, 3, 0))))        # TILED, other permutation
p2c = S.make_plan(lambda x: x, None, None, A.size, (C, A.permutedims((3, 2, 1, 0))))        # TILED, same permutation, other destination
p3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in perms))
p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
alone = {k: t(lambda p=p: p.execute(cur())) for k, p in (("perm", p2), ("perm2341", p2b), ("permC", p2c), ("sum4", p3), ("copy", pc))}
print("alone:", {k: round(v, 2) for k, v in alone.items()})
plans = {"perm": p2, "perm2341": p2b, "permC": p2c, "sum4": p3, "copy": pc}
for x, y in (("perm", "permC"), ("perm", "perm2341"), ("perm", "copy"), ("sum4", "copy"), ("perm", "sum4"), ("sum4", "sum4")):
    def pair(x=x, y=y):
        plans[x].execute(cur())
        plans[y].execute(cur())
    tp = t(pair)
    print(f"{x:8s} + {y:8s}: pair {tp:.2f} us, alone sum {alone[x] + alone[y]:.2f}, penalty {tp - alone[x] - alone[y]:+.2f}")
# a tiny kernel (16 elements) in between: is the penalty about data in the caches or about switching kernels?
tS = torch.randn(16, dtype=torch.float64, device="cuda")
tT = torch.empty_like(tS)
tiny = S.make_plan(lambda x: x, None, None, (16,), (S.StridedView(tT, (16,), (1,), 0), S.StridedView(tS, (16,), (1,), 0)))
plans["tiny"] = tiny
alone["tiny"] = t(lambda: tiny.execute(cur()))
print("tiny alone", round(alone["tiny"], 2), tiny.describe()[:60])
for x in ("sum4", "perm", "copy"):
    def pair(x=x):
        plans[x].execute(cur())
        tiny.execute(cur())
    tp = t(pair)
    print(f"{x:8s} + tiny    : pair {tp:.2f} us, alone sum {alone[x] + alone['tiny']:.2f}, penalty {tp - alone[x] - alone['tiny']:+.2f}")
