import sys, os
sys.path.insert(0, os.getcwd())
import torch
import strided_jl_amd as S
from bench import colmajor_view, event_time_ms, graph_of
def cur(): return int(torch.cuda.current_stream().cuda_stream)
def time_plan(plan, reps=300):
    plan.execute(cur()); torch.cuda.synchronize()
    g = graph_of(torch, lambda: plan.execute(cur()), reps); g.replay(); torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 3) for _ in range(3)) / reps * 1e3
n=32
ts=[torch.randn(n**4, dtype=torch.float64, device="cuda") for _ in range(5)]
V=[colmajor_view(S,t,(n,)*4) for t in ts]
perms=[(0,1,2,3),(1,2,3,0),(2,3,0,1),(3,0,1,2)]
f4=lambda a,b,c,d:a+b+c+d
rows=[("stream add4, 4 distinct arrays", f4, (V[0],V[1],V[2],V[3],V[4])),
      ("stream add2, 2 distinct arrays", lambda a,b:a+b, (V[0],V[1],V[2])),
      ("stream add3, 3 distinct", lambda a,b,c:a+b+c, (V[0],V[1],V[2],V[3])),
      ("tiled add4, 4 distinct arrays permuted", f4, (V[0],)+tuple(V[1+i].permutedims(q) for i,q in enumerate(perms))),
      ("tiled add4, one array 4 perms (C3)", f4, (V[0],)+tuple(V[1].permutedims(q) for q in perms)),
      ("tiled add2: A + perm(A)", lambda a,b:a+b, (V[0],V[1],V[1].permutedims(perms[1]))),
      ("tiled add2: A + perm2(A)", lambda a,b:a+b, (V[0],V[1],V[1].permutedims(perms[2]))),
      ("tiled add3: A + 2 perms", lambda a,b,c:a+b+c, (V[0],V[1],V[1].permutedims(perms[1]),V[1].permutedims(perms[2]))),
      ("tiled ident perm1", lambda a:a, (V[0],V[1].permutedims(perms[1]))),
      ]
for name,f,arrs in rows:
    p=S.make_plan(f,None,None,arrs[0].size,arrs)
    us=time_plan(p)
    d=p.describe()
    print(f"{name:42s} {us:7.2f} us | {d[:d.find(' algbytes')]}")
