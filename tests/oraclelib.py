"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference algorithm.
TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

SMR_MAXN, SMR_MAXM = 8, 8


class oracle_plan_info(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("M", C.c_int32), ("g", C.c_int32), ("_pad", C.c_int32),
        ("fused", C.c_int64 * SMR_MAXN),
        ("importance", C.c_int64 * SMR_MAXN),
        ("perm", C.c_int32 * SMR_MAXN),
        ("dims", C.c_int64 * SMR_MAXN),
        ("strides", (C.c_int64 * SMR_MAXN) * SMR_MAXM),
        ("costs", C.c_int64 * SMR_MAXN),
        ("blocks", C.c_int64 * SMR_MAXN),
    ]


def ensure_built():
    src = os.path.join(ORACLE_DIR, "strided_oracle.cpp")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        r = subprocess.run(["make", "-C", ORACLE_DIR], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building liboracle.so failed:\n" + r.stdout + r.stderr)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        ensure_built()
        _lib = C.CDLL(LIB)
        _lib.oracle_last_error.restype = C.c_char_p
        _lib.oracle_mapreduce.argtypes = [C.c_void_p, C.c_int]
        _lib.oracle_plan.argtypes = [C.c_void_p, C.POINTER(oracle_plan_info)]
        _lib.oracle_threaded_boxes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
        _lib.oracle_indexorder.argtypes = [C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64)]
    return _lib


def mapreduce(problem, nthreads=1):
    rc = load().oracle_mapreduce(C.byref(problem), nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}: {load().oracle_last_error().decode()}")


def set_literal_409(on):
    """src/mapreduce.jl:409 read literally (`stride > 0`): initop is skipped for later blocks along a reversed kept dim of the destination"""
    load().oracle_set_literal_409(1 if on else 0)


def guard_hits(reset=True):
    """How often the termination guard of the restated _computeblocks fired (oracle/strided_oracle.cpp: with negative strides the
    reference's halving loops are suspected never to end; the oracle breaks out)."""
    lib = load()
    lib.oracle_guard_hits.restype = C.c_long
    return int(lib.oracle_guard_hits(1 if reset else 0))


def plan(problem):
    info = oracle_plan_info()
    rc = load().oracle_plan(C.byref(problem), C.byref(info))
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}: {load().oracle_last_error().decode()}")
    N, M = info.N, info.M
    return dict(
        N=N, M=M, g=info.g,
        fused=tuple(info.fused[:N]), importance=tuple(info.importance[:N]),
        perm=tuple(info.perm[:N]), dims=tuple(info.dims[:N]),
        strides=tuple(tuple(info.strides[k][:N]) for k in range(M)),
        costs=tuple(info.costs[:N]), blocks=tuple(info.blocks[:N]))


def threaded_boxes(problem, nthreads):
    N, M = problem.N, problem.M
    cap = 4096
    buf = (C.c_int64 * (cap * (N + M)))()
    n = load().oracle_threaded_boxes(C.byref(problem), nthreads, buf, cap)
    out = []
    for i in range(min(n, cap)):
        row = buf[i * (N + M):(i + 1) * (N + M)]
        out.append((tuple(row[:N]), tuple(row[N:])))
    return out


def indexorder(strides):
    n = len(strides)
    a = (C.c_int64 * n)(*strides)
    o = (C.c_int64 * n)()
    load().oracle_indexorder(a, n, o)
    return tuple(o)
