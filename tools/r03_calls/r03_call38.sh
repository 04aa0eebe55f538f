#!/bin/bash
# kernel-trace of the two-sided FLAT form and the floor-less ROW reduction (durations next to the HIP-event numbers of the A/B tools)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/c38
mkdir -p $O
cat > /tmp/kt_new.py <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import strided_jl_amd as S
from bench import colmajor_view
cur = lambda: int(torch.cuda.current_stream().cuda_stream)
def perm(shape, q, dt):
    n = int(np.prod(shape)); tA = torch.randn(n, dtype=dt, device="cuda"); tB = torch.empty_like(tA)
    A = colmajor_view(S, tA, shape); B = colmajor_view(S, tB, tuple(shape[i] for i in q))
    return S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(q))), (tA, tB)
def red(dims, rd, dt):
    n = int(np.prod(dims)); tA = torch.randn(n, dtype=dt, device="cuda"); A = colmajor_view(S, tA, dims)
    out = A.similar(size=tuple(1 if d in rd else m for d, m in enumerate(dims)))
    return S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, A)), (tA, out)
plans = [perm((5, 300, 300, 7), (3, 2, 1, 0), torch.float64), perm((6, 64, 64, 64, 5), (4, 3, 2, 1, 0), torch.float64), perm((5, 300, 300, 7), (3, 1, 2, 0), torch.float64),
         red((100, 90, 80, 7), (0,), torch.float32), red((3, 1920, 1080), (0,), torch.float32), red((512, 384, 64), (1,), torch.float32)]
for p, keep in plans:
    print(p.describe())
    for _ in range(300):
        p.execute(cur())
    torch.cuda.synchronize()
PY
( cd /tmp && export TMPDIR=/tmp && timeout 70 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python /tmp/kt_new.py > $R/$O/kt.log 2>&1 )
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/new_kernels_trace_stats.txt 2>&1
grep -v "^$" $O/kt.log | grep "family=" >> $O/new_kernels_trace_stats.txt
head -14 $O/new_kernels_trace_stats.txt | cut -c1-200
rm -rf $O/kt
