"""Is it the ragged last tile or the row stride?  Transposes into a 96 x 7200 / 100 x 7200 destination that sits inside parents of 96 .. 128 rows
(row strides of 768 .. 1024 bytes): whole tiles with misaligned rows against ragged tiles with aligned rows.  Usage: python tools/row_stride_cases.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import strided_jl_amd as S
from bench import colmajor_view, event_time_ms, graph_of
cur = lambda: int(torch.cuda.current_stream().cuda_stream)
def mk(dims, dt=torch.float64):
    return colmajor_view(S, torch.randn(int(np.prod(dims)), dtype=dt, device="cuda"), dims)
def t(plan, reps=40):
    plan.execute(cur()); torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps); g.replay(); torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3
A = mk((7200, 96))
for ld in (96, 100, 104, 112, 128):
    P = mk((ld, 7200))
    B = P.sview(slice(0, 96), slice(None))
    plan = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims((1, 0))))
    d = plan.describe()
    print("dest 96 x 7200 inside a parent of %3d rows (row stride %4d B): %5.2f us | %s" % (ld, ld * 8, t(plan), d[d.find("family="):d.find(" ct=")] + d[d.find(" dims="):d.find(" algbytes")][:70]))
A2 = mk((7200, 100))
for ld in (100, 128):
    P = mk((ld, 7200))
    B = P.sview(slice(0, 100), slice(None))
    plan = S.make_plan(lambda x: x, None, None, B.size, (B, A2.permutedims((1, 0))))
    d = plan.describe()
    print("dest 100 x 7200 inside a parent of %3d rows (row stride %4d B): %5.2f us | %s" % (ld, ld * 8, t(plan), d[d.find("family="):d.find(" ct=")] + d[d.find(" dims="):d.find(" algbytes")][:70]))
