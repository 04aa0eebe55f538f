#!/usr/bin/env python3
"""Host-side concurrency stress (no GPU): N threads plan, describe, canonicalise, generate functor source and analyse sequences at the
same time through the C ABI (ctypes releases the GIL inside every call).  Meant to run against the ThreadSanitizer build of the host
side: SMR_LIB=/tmp/smr_tsan/libstrided_hip_tsan.so LD_PRELOAD=<libclang_rt.tsan> python tools/thread_stress.py [threads] [iterations]"""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import strided_jl_amd as S  # noqa: E402

fn = S.fn
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 8
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = S._lib.load()
errors = []


def view(shape, dt, perm=None):
    st, s = [], 1
    for d in shape:
        st.append(s)
        s *= d
    v = S.StridedView(np.zeros(s, dtype=dt), tuple(shape), tuple(st), 0)   # (real footprints: the sequence analysis compares byte ranges)
    return v.permutedims(perm) if perm else v


def work(tid):
    rng = np.random.default_rng(tid)
    try:
        for it in range(IT):
            n = int(rng.integers(3, 40))
            dt = [np.float32, np.float64, np.complex64, np.int32, np.int64][int(rng.integers(0, 5))]
            A = view((n, n + 1, n + 2), dt)
            P = view((n + 2, n + 1, n), dt, (2, 1, 0))
            kind = it % 5
            if kind == 0:
                p = S.make_plan(lambda x: x, None, None, A.size, (A, P))
                assert "family=" in p.describe()
            elif kind == 1:
                B = view((n, n + 1, n + 2), dt)
                p = S.make_plan(lambda x, y: x * y + x, None, None, A.size, (A, B, P))
                p.jit_source()
            elif kind == 2:
                r = S.StridedView(np.zeros(1, dtype=dt), A.size, (0, 0, 0), 0)
                S.make_plan(fn.abs2 if dt not in (np.int32, np.int64) else (lambda x: x * x), "+", None, A.size, (r, A)).describe()
            elif kind == 3:
                try:   # an error path: the error string is per thread
                    S.make_plan(lambda x: x / 3, None, None, A.size, (view(A.size, np.float64), view(A.size, np.int64)))
                except S._lib.UnsupportedOnDevice as e:
                    assert "64-bit integer" in str(e), str(e)
                S.set_option("reduce_part_wgs", 1024)
                assert S.get_option("reduce_part_wgs") == 1024
            else:
                q = S.Sequence()
                B = view((n, n + 1, n + 2), dt)
                q.add(S.make_plan(lambda x: x, None, None, A.size, (A, P)))
                q.add(S.make_plan(lambda x: x + 1, None, None, A.size, (B, A)))
                assert q.components() == [0, 0] and q.fences()[0] == [0, 1]   # the second execution reads what the first writes
                del q
    except Exception as e:  # noqa: BLE001
        import traceback
        errors.append((tid, traceback.format_exc()[-600:]))


ths = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
for t in ths:
    t.start()
for t in ths:
    t.join()
print("%d threads x %d iterations: %s" % (NT, IT, "ok" if not errors else errors[:3]))
sys.exit(1 if errors else 0)
