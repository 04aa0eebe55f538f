import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import strided_jl_amd as S
from bench import colmajor_view
lib = S._lib.load()
dims = (100, 90, 80, 7)
dt = torch.float32 if sys.argv[2] == "f32" else torch.float64
A = colmajor_view(S, torch.randn(int(np.prod(dims)), dtype=dt, device="cuda"), dims)
cur = lambda: int(torch.cuda.current_stream().cuda_stream)
rd = tuple(int(x) for x in sys.argv[1].split(","))
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    S._lib.check(lib.smr_set_option(k.encode(), int(v)))
out = A.similar(size=tuple(1 if d in rd else m for d, m in enumerate(dims)))
plan = S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A))
print(rd, plan.describe())
for _ in range(200):
    plan.execute(cur())
torch.cuda.synchronize()
