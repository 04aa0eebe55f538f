#!/usr/bin/env python3
"""Profiler-independent device-side timing of the headline kernels.

The stamp build of the library (`make -C strided.jl_amd/csrc stamp` -> libstrided_hip_stamp.so, -DSMR_STAMP=1) makes
every wave of the TILED / ORBIT / STREAM kernels record s_memrealtime (one 100 MHz clock for the whole device) at
entry and after its last store was acknowledged.  Here R launches are captured into one hipGraph, every launch
with its own stamp region, the graph is replayed, and per launch we report
    span     = last wave end - first wave start              (what a perfect kernel trace would call the duration)
    cadence  = first start of launch i+1 - first start of i   (what back-to-back HIP events measure per launch)
    gap      = cadence - span                                 (kernel boundary: drain, cache write-back, dispatch)
next to the HIP-event time of the same (stamped) graph and of the product library (re-run with --plain).

Usage: python tools/device_span.py [--n 32] [--reps 200]
"""
import argparse
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP_LIB = os.path.join(ROOT, "strided.jl_amd", "libstrided_hip_stamp.so")

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--plain", action="store_true", help="product library, HIP events only (called by the stamped run)")
args = ap.parse_args()

if not args.plain and os.environ.get("SMR_LIB") != STAMP_LIB:
    if not os.path.exists(STAMP_LIB):
        sys.exit("build the stamp library first: make -C strided.jl_amd/csrc stamp")
    env = dict(os.environ, SMR_LIB=STAMP_LIB)
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms, graph_of  # noqa: E402

lib = S._lib.load()
n, R = args.n, args.reps
dev = torch.device("cuda", 0)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
tB = torch.empty_like(tA)
tC = torch.empty_like(tA)
A, B, Cc = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
plans = {
    "permutedims!(4,3,2,1)": S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0)))),
    "4-way permuted sum": S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms)),
    "contiguous copy": S.make_plan(lambda x: x, None, None, A.size, (B, A)),
}


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def events_us(fn, reps):
    g = graph_of(torch, fn, reps)
    g.replay()
    torch.cuda.synchronize()
    return min(event_time_ms(torch, g.replay, 2) for _ in range(7)) / reps * 1e3, g


if args.plain:
    for name, p in plans.items():
        us, _ = events_us(lambda p=p: p.execute(cur()), R)
        print("PLAIN\t%s\t%.3f" % (name, us))
    p2, p3 = plans["permutedims!(4,3,2,1)"], plans["4-way permuted sum"]
    us, _ = events_us(lambda: (p2.execute(cur()), p3.execute(cur())), R)
    print("PLAIN\tstep (both, in order)\t%.3f" % us)
    sys.exit(0)

assert lib.smr_get_option(b"stamp_build") == 1, "not the stamp build"
hip = C.CDLL("libamdhip64.so")
khz = C.c_int(0)
hip.hipDeviceGetAttribute(C.byref(khz), 10017, 0)  # hipDeviceAttributeWallClockRate
tick_us = 1e3 / khz.value if khz.value > 0 else 0.01
print("device wall clock: %d kHz (%.1f ns per tick); %s; n = %d, %d launches per graph" % (khz.value, tick_us * 1e3, torch.cuda.get_device_name(0), n, R))

plain = {}
r = subprocess.run([sys.executable, os.path.abspath(__file__), "--plain", "--n", str(n), "--reps", str(R)], capture_output=True, text=True,
                   env={k: v for k, v in os.environ.items() if k != "SMR_LIB"})
for line in r.stdout.splitlines():
    if line.startswith("PLAIN\t"):
        _, name, us = line.split("\t")
        plain[name] = float(us)

words = 64 << 20
stamps = torch.zeros(words, dtype=torch.int64, device=dev)
S._lib.check(lib.smr_set_option(b"stamp_base", stamps.data_ptr()))
S._lib.check(lib.smr_set_option(b"stamp_cap", words))


def analyse(name, fns, labels):
    """fns: the launches of ONE repetition (1 kernel, or the 2 kernels of the step)."""
    S._lib.check(lib.smr_set_option(b"stamp_used", 0))
    marks = []

    def rep():
        for f in fns:
            before = lib.smr_get_option(b"stamp_used")
            f()
            marks.append((before, lib.smr_get_option(b"stamp_used")))

    # graph_of runs fn once eagerly (warm-up) before capturing R repetitions: regions of the warm-up are dropped
    us, g = events_us(rep, R)
    regions = marks[len(fns):len(fns) * (R + 1)]
    stamps.zero_()
    torch.cuda.synchronize()
    for _ in range(6):   # back to back (warm clocks, like the event-timed replays); the last replay's stamps remain
        g.replay()
    torch.cuda.synchronize()
    h = stamps[: regions[-1][1]].cpu().numpy()
    first, last, ramp50, ramp95, rampmax, life, nw = [], [], [], [], [], [], []
    for (a, b) in regions:
        seg = h[a:b].reshape(-1, 2)
        seg = seg[seg[:, 0] != 0]
        if len(seg) == 0:
            sys.exit("no stamps in a region: is this kernel instrumented? (%s)" % name)
        f0 = seg[:, 0].min()
        first.append(int(f0))
        last.append(int(seg[:, 1].max()))
        st = sorted(seg[:, 0] - f0)
        ramp50.append(st[len(st) // 2])
        ramp95.append(st[len(st) * 95 // 100])
        rampmax.append(st[-1])
        life.append(float((seg[:, 1] - seg[:, 0]).mean()))
        nw.append(len(seg))
    k = len(fns)
    med = statistics.median
    print("%s: HIP events %.3f us per repetition in the stamped build (product library: %s us)" % (name, us, "%.3f" % plain[name] if name in plain else "?"))
    skip = 10 * k
    for j in range(k):
        idx = [i for i in range(skip + j, len(regions) - k, k)]
        span = med([(last[i] - first[i]) * tick_us for i in idx])
        nxt = med([(first[i + 1] - first[i]) * tick_us for i in idx])
        gap = med([(first[i + 1] - last[i]) * tick_us for i in idx])
        print("    %-24s span %6.2f us | to next kernel's first wave %6.2f us (gap after last wave %5.2f) | wave starts after the first: p50 %4.2f p95 %4.2f "
              "max %4.2f us | mean wave lifetime %5.2f us | %d waves" % (labels[j], span, nxt, gap, med([ramp50[i] for i in idx]) * tick_us,
                                                                          med([ramp95[i] for i in idx]) * tick_us, med([rampmax[i] for i in idx]) * tick_us,
                                                                          med([life[i] for i in idx]) * tick_us, nw[idx[0]]))
    idx = list(range(skip, len(regions) - k, k))
    cad = med([(first[i + k] - first[i]) * tick_us for i in idx])
    print("    device cadence per repetition %.3f us  (events, stamped build: %.3f; product library: %s)" % (cad, us, "%.3f" % plain[name] if name in plain else "?"))
    return cad


for name, p in plans.items():
    print(p.describe())
    analyse(name, [lambda p=p: p.execute(cur())], [name])
p2, p3 = plans["permutedims!(4,3,2,1)"], plans["4-way permuted sum"]
analyse("step (both, in order)", [lambda: p2.execute(cur()), lambda: p3.execute(cur())], ["permutedims!", "4-way sum"])
