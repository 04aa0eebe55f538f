#!/usr/bin/env python3
"""Round 5: the per-launch times of the two headline kernels, every witness in one file.

  HIP events      back-to-back launches of ONE kernel inside a hipGraph (what bench.py's roofline object reports): span + HIP's kernel
                  boundary (agent-scope acquire + release by the packet processor, ~1.8 us)
  device stamps   stamp build (tools/device_span.py): first wave start -> last wave end, and start-to-start cadence
  library replay  the same kernel replayed alone by smr_seq (its own AQL packets: self-released launch, no fences inside the replay;
                  one queue, and cut in two on two queues), host wall clock / launches over 2000 launches
  rocprofv3       --kernel-trace --stats average of the profiled bench run (HIP launch path: the direct path switches itself off under
                  a profiler), read from the file given with --rocprof
Algorithmic bytes per launch: 16,777,216 (32^4 Float64: 8 MiB in + 8 MiB out); frac = of 8000 GB/s."""
import argparse
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--rocprof", default=None, help="a *_bench_kernel_trace_stats.txt file")
args = ap.parse_args()
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

n = 32
dev = torch.device("cuda", 0)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
tB, tC = torch.empty_like(tA), torch.empty_like(tA)
A, B, Cc = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
plans = {"permutedims!(4,3,2,1)": S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0)))),
         "4-way permuted sum": S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))}
BYTES = 2 * 8 * n ** 4


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


rows = {k: {} for k in plans}
for name, p in plans.items():
    g = graph_of(torch, lambda p=p: p.execute(cur()), 500)
    g.replay()
    torch.cuda.synchronize()
    rows[name]["HIP events, hipGraph of 500 launches (min of 9)"] = min(event_time_ms(torch, g.replay, 2) for _ in range(9)) / 500 * 1e3
    st = S.Stream()
    for label, opts in (("library replay, one queue", {"queues": 1, "slices": 1}), ("library replay, cut in two (2 queues)", {"queues": 2, "slices": 2})):
        q = S.Sequence().add(p)
        for k, v in opts.items():
            q.set(k, v)
        q.run(50, st.handle); q.wait()
        best = 1e30
        for _ in range(7):
            torch.cuda.synchronize()
            t = time.perf_counter()
            q.run(2000, st.handle); q.wait()
            best = min(best, time.perf_counter() - t)
        rows[name][label] = best / 2000 * 1e6
        del q
    st.close()
# device stamps (stamp build, separate process)
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "device_span.py"), "--n", "32", "--reps", "200"], capture_output=True, text=True)
span_txt = r.stdout
for name in plans:
    m = re.search(r"^\s+" + re.escape(name) + r"\s+span\s+([0-9.]+) us \| to next kernel's first wave\s+([0-9.]+) us", span_txt, re.M)
    if m:
        rows[name]["device stamps: span (first wave start -> last wave end), stamp build"] = float(m.group(1))
        rows[name]["device stamps: start-to-start cadence, stamp build (graph replay through HIP)"] = float(m.group(2))
if args.rocprof and os.path.exists(args.rocprof):
    for line in open(args.rocprof):
        f = line.split()
        if len(f) >= 5 and f[0].isdigit():
            if "k_orbit_" in line and "FAdd4" in line:
                rows["4-way permuted sum"]["rocprofv3 --kernel-trace --stats average (HIP launch path), %s" % os.path.basename(args.rocprof)] = float(f[1]) / 1e3
            if "k_tiled_map" in line and "FIdent" in line:
                rows["permutedims!(4,3,2,1)"]["rocprofv3 --kernel-trace --stats average (HIP launch path), %s" % os.path.basename(args.rocprof)] = float(f[1]) / 1e3
print("%s; 32^4 Float64, algorithmic bytes per launch %d" % (torch.cuda.get_device_name(0), BYTES))
for name, d in rows.items():
    print("\n%s   [%s]" % (name, plans[name].describe()))
    for label, us in d.items():
        print("   %-92s %7.3f us  %7.0f GB/s  frac %.3f" % (label, us, BYTES / us / 1e3, BYTES / us / 1e3 / 8000))
print("\n---- raw output of tools/device_span.py ----")
print(span_txt)
