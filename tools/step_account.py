#!/usr/bin/env python3
"""Round 5: a device-side busy / idle account of the replayed bench step.

Stamp build of the library (make -C strided.jl_amd/csrc stamp): every wave of the TILED / ORBIT kernels records the device's 100 MHz
wall clock at entry and after its last store was acknowledged.  The bench step (32^4 Float64: permutedims!(B, A, (4,3,2,1)) and
C .= sum of 4 permuted views of A) is recorded NS times in a row as separate executions -- every launch then owns a stamp region --
and replayed once by smr_seq in each dispatch form; per form and per steady-state step (the first 6 and last 4 are dropped):
    span        first wave start -> last wave end of a launch (or of a slice of it)
    residency   time with waves of BOTH operations on the device / of one only / of none (nothing of this process resident)
    period      start-to-start of consecutive steps = what the replay delivers per step
and, against the host clock of the same replay: wall time of smr_seq_run + smr_seq_wait minus the device span = the fixed cost
(doorbell -> first wave, last wave -> completion observed).
The stamped kernels are 0.15-0.25 us per launch slower than the product build (two clock reads, one wait, one 16-byte store per wave).
"""
import ctypes as C
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP_LIB = os.path.join(ROOT, "strided.jl_amd", "libstrided_hip_stamp.so")
if os.environ.get("SMR_LIB") != STAMP_LIB:
    if not os.path.exists(STAMP_LIB):
        sys.exit("build the stamp library first: make -C strided.jl_amd/csrc stamp")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, SMR_LIB=STAMP_LIB)))

sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view  # noqa: E402

lib = S._lib.load()
assert lib.smr_get_option(b"stamp_build") == 1, "not the stamp build"
n, NS = 32, 30
dev = torch.device("cuda", 0)
tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
tB, tC = torch.empty_like(tA), torch.empty_like(tA)
A, B, Cc = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
p3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in perms))
hip = C.CDLL("libamdhip64.so")
khz = C.c_int(0)
hip.hipDeviceGetAttribute(C.byref(khz), 10017, 0)  # hipDeviceAttributeWallClockRate
tick = 1e3 / khz.value if khz.value > 0 else 0.01   # us per tick
words = 32 << 20
stamps = torch.zeros(words, dtype=torch.int64, device=dev)
S._lib.check(lib.smr_set_option(b"stamp_base", stamps.data_ptr()))
S._lib.check(lib.smr_set_option(b"stamp_cap", words))
st = S.Stream()
print("%s, device wall clock %d kHz; 32^4 Float64, %d recorded steps per replay; stamped build" % (torch.cuda.get_device_name(0), khz.value, NS))


def used():
    return lib.smr_get_option(b"stamp_used")


def record(items, opts):
    """-> (sequence, [(kind, lo, hi)] stamp regions in item order)"""
    S._lib.check(lib.smr_set_option(b"stamp_used", 0))
    opts = dict(opts)
    self_rel = opts.pop("_seq_self_release", 1)   # library option (not a sequence setting): write-through stores for recorded launches
    S.set_option("seq_self_release", self_rel)
    q = S.Sequence()
    for it in items:
        q.add(it)
    for k, v in opts.items():
        q.set(k, v)
    q.info()                       # builds: every launch is recorded once, in item order, and takes its region
    S.set_option("seq_self_release", 1)
    total = used()
    per = {}
    # region sizes: a one-item sequence of each plan
    for name, p in (("perm", p2), ("sum", p3)):
        S._lib.check(lib.smr_set_option(b"stamp_used", 0))
        t = S.Sequence().add(p)
        t.set("slices", 1)
        t.info()
        per[name] = used()
        del t
    regs, at = [], 0
    for it in items:
        name = "perm" if it is p2 else "sum"
        regs.append((name, at, at + per[name]))
        at += per[name]
    assert at == total, (at, total)
    S._lib.check(lib.smr_set_option(b"stamp_used", total))
    return q, regs


def spans(h, regs):
    out = []
    for name, lo, hi in regs:
        seg = h[lo:hi].reshape(-1, 2)
        seg = seg[seg[:, 0] != 0]
        if len(seg) == 0:
            sys.exit("no stamps in a region (%s)" % name)
        out.append((name, int(seg[:, 0].min()), int(seg[:, 1].max()), seg))
    return out


def union_len(iv):
    iv = sorted(iv)
    tot, cur_lo, cur_hi = 0, None, None
    for lo, hi in iv:
        if cur_hi is None or lo > cur_hi:
            if cur_hi is not None:
                tot += cur_hi - cur_lo
            cur_lo, cur_hi = lo, hi
        else:
            cur_hi = max(cur_hi, hi)
    if cur_hi is not None:
        tot += cur_hi - cur_lo
    return tot


def account(title, items, opts):
    q, regs = record(items, opts)
    for _ in range(4):
        q.run(1, st.handle); q.wait()
    stamps.zero_()
    torch.cuda.synchronize()
    walls = []
    for _ in range(5):             # the last replay's stamps remain
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q.run(1, st.handle); q.wait()
        walls.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    h = stamps[: regs[-1][2]].cpu().numpy()
    sp = spans(h, regs)
    info = q.info()
    nper = 2 if any(r[0] == "perm" for r in regs) and any(r[0] == "sum" for r in regs) else 1
    nsteps = len(items) // nper
    first = min(s[1] for s in sp)
    last = max(s[2] for s in sp)
    dev_span = (last - first) * tick
    print("\n== %s ==\n   %s" % (title, " ".join(w for w in info.split() if w.split("=")[0] in ("queues", "sliced", "packets", "acquire", "first_acquire", "release", "self_released"))))
    print("   replay of %d steps: host wall clock %.2f us (best of 5: %.2f), device first wave -> last wave %.2f us  => fixed cost %.2f us; %.3f us per step over the whole replay"
          % (nsteps, walls[-1], min(walls), dev_span, walls[-1] - dev_span, dev_span / nsteps))
    med = statistics.median
    steady = range(6, nsteps - 4)
    rows = {}
    for name in ("perm", "sum"):
        idx = [i for i, s in enumerate(sp) if s[0] == name]
        if not idx:
            continue
        sel = [idx[k] for k in steady]
        rows[name] = dict(span=med([(sp[i][2] - sp[i][1]) * tick for i in sel]),
                          period=med([(sp[idx[k + 1]][1] - sp[idx[k]][1]) * tick for k in steady]),
                          gap=med([(sp[idx[k + 1]][1] - sp[idx[k]][2]) * tick for k in steady]),
                          life=med([float((sp[i][3][:, 1] - sp[i][3][:, 0]).mean()) * tick for i in sel]))
        print("   %-4s launch: span %.2f us (first wave start -> last wave end, all slices), start-to-start %.2f us, end -> next start %+.2f us, mean wave lifetime %.2f us"
              % (name, rows[name]["span"], rows[name]["period"], rows[name]["gap"], rows[name]["life"]))
    # residency over the steady window: intervals of launches (kernel-level: first wave start .. last wave end)
    lo_t = min(sp[i][1] for i, s in enumerate(sp) if (i // nper) == steady[0])
    hi_t = min(sp[i][1] for i, s in enumerate(sp) if (i // nper) == steady[-1] + 1)
    clip = lambda a, b: (max(a, lo_t), min(b, hi_t))  # noqa: E731
    iv = {name: [clip(s[1], s[2]) for s in sp if s[0] == name and s[2] > lo_t and s[1] < hi_t] for name in ("perm", "sum")}
    window = (hi_t - lo_t) * tick
    nst = len(steady)
    pl, sl = union_len(iv["perm"]) * tick, union_len(iv["sum"]) * tick
    any_l = union_len(iv["perm"] + iv["sum"]) * tick
    both = pl + sl - any_l
    # wave-level: fraction of the window in which at least one wave of this process is resident
    wiv = []
    for s in sp:
        if s[2] > lo_t and s[1] < hi_t:
            seg = s[3]
            wiv += [clip(int(a), int(b)) for a, b in seg if b > lo_t and a < hi_t]
    wave_any = union_len(wiv) * tick
    print("   per step over %d steady steps (window %.2f us = %.3f us per step):" % (nst, window, window / nst))
    print("      both operations resident %.2f us | only permutedims! %.2f us | only the sum %.2f us | no launch resident %.2f us  (kernel level: first wave start .. last wave end)"
          % (both / nst, (pl - both) / nst, (sl - both) / nst, (window - any_l) / nst))
    print("      at least one wave of this process resident: %.2f us of %.3f us per step = %.1f %% device-busy" % (wave_any / nst, window / nst, 100 * wave_any / window))
    del q
    return dict(per_step=window / nst, busy=wave_any / window, rows=rows)


step = [p2, p3] * NS
res = {}
res["default"] = account("library default: one queue per dependency component, every chain cut in two (4 queues), self-released launches, acquire by need", step, {})
res["q3"] = account("3 queues: permutedims! | sum/2 | sum/2", step, {"queues": 3, "slices:1": 2})
res["q2"] = account("2 queues: one per dependency component, nothing cut", step, {"queues": 2, "slices": 1})
res["q2_r04"] = account("2 queues, round-4 fences (agent-scope acquire + release on every packet, plain / non-temporal stores)", step,
                        {"queues": 2, "slices": 1, "acquire": 1, "_seq_self_release": 0})
res["q1"] = account("1 queue, in recorded order", step, {"queues": 1})
res["perm_alone"] = account("permutedims! alone, one chain", [p2] * NS, {"queues": 1, "slices": 1})
res["sum_alone"] = account("the 4-way sum alone, one chain", [p3] * NS, {"queues": 1, "slices": 1})
res["sum_alone2"] = account("the 4-way sum alone, cut in two (2 queues)", [p3] * NS, {"queues": 2, "slices": 2})
print("\nsummary (us per step, stamped build): " + ", ".join("%s %.2f (%.0f %% busy)" % (k, v["per_step"], 100 * v["busy"]) for k, v in res.items()))
st.close()
