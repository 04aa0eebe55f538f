"""ctypes binding of libstrided_hip.so -- the C ABI declared in include/strided_hip.h.

The HIP library is the only compute path of this package: if it is missing or no gfx950
device is usable, every compute entry point raises (there is no CPU fallback; the CPU
restatement under oracle/ is test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

SMR_MAXN = 8
SMR_MAXM = 8
SMR_MAXPROG = 96
SMR_MAXCONST = 16

# smr_status
SMR_OK, SMR_EINVAL, SMR_EUNSUPPORTED, SMR_EHIP, SMR_ENOMEM, SMR_ENODEVICE = 0, -1, -2, -3, -4, -5

# smr_dtype
(SMR_F32, SMR_F64, SMR_C32, SMR_C64, SMR_I8, SMR_I16, SMR_I32, SMR_I64,
 SMR_U8, SMR_U16, SMR_U32, SMR_U64, SMR_BOOL) = range(13)

# smr_redop / smr_initop
SMR_RED_NONE, SMR_RED_ADD, SMR_RED_MUL, SMR_RED_MIN, SMR_RED_MAX, SMR_RED_AND, SMR_RED_OR = range(7)
(SMR_INIT_NONE, SMR_INIT_IDENTITY, SMR_INIT_ZERO, SMR_INIT_SCALE, SMR_INIT_CONST,
 SMR_INIT_CONJ) = range(6)

# smr_opcode
OPCODES = dict(
    ARG=0, CONST=1,
    NEG=8, ABS=9, ABS2=10, CONJ=11, REAL=12, IMAG=13, SQRT=14, EXP=15, LOG=16, SIN=17, COS=18,
    TANH=19, INV=20, ROUND32=21, WIDEN=22,
    ADD=32, SUB=33, MUL=34, DIV=35, MIN=36, MAX=37, LT=38, LE=39, GT=40, GE=41, EQ=42, NE=43,
    SELECT=64,
)


class smr_operand(C.Structure):
    _fields_ = [
        ("base", C.c_void_p),
        ("offset", C.c_int64),
        ("strides", C.c_int64 * SMR_MAXN),
        ("dtype", C.c_int32),
        ("conj", C.c_int32),
    ]


class smr_problem(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("M", C.c_int32),
        ("dims", C.c_int64 * SMR_MAXN),
        ("ops", smr_operand * SMR_MAXM),
        ("fprog", C.POINTER(C.c_uint8)),
        ("fprog_len", C.c_int32),
        ("nconsts", C.c_int32),
        ("fconsts", C.POINTER(C.c_double)),
        ("redop", C.c_int32),
        ("initop", C.c_int32),
        ("initarg", C.c_double * 2),
        ("stream", C.c_void_p),
    ]


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# $SMR_LIB: another build of the library (A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get("SMR_LIB") or os.path.join(_PKG_DIR, "libstrided_hip.so")
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")

# every symbol include/strided_hip.h declares
EXPORTS = [
    "smr_abi_version", "smr_init", "smr_shutdown", "smr_device_count", "smr_last_error",
    "smr_malloc", "smr_free", "smr_memcpy_h2d", "smr_memcpy_d2h", "smr_stream_sync",
    "smr_mapreduce", "smr_plan_create", "smr_plan_execute", "smr_plan_destroy",
    "smr_plan_describe", "smr_plan_algorithmic_bytes", "smr_plan_tile_order", "smr_plan_orbit_pairs", "smr_plan_flat_runs", "smr_plan_flat_side", "smr_plan_flat_batched", "smr_mapreduce_scalar", "smr_plan_jit_compile", "smr_plan_jit_source", "smr_plan_prepare", "smr_comm_unique_id", "smr_comm_init", "smr_comm_rank", "smr_comm_library",
    "smr_comm_destroy", "smr_mapreduce_sharded", "smr_mapreduce_sharded_ex", "smr_shard", "smr_shard_ex", "smr_init_reduction", "smr_set_option",
    "smr_get_option", "smr_overlap_begin", "smr_overlap_end", "smr_overlap_fence", "smr_stream_create", "smr_stream_destroy",
    "smr_seq_create", "smr_seq_add", "smr_seq_run", "smr_seq_wait", "smr_seq_info", "smr_seq_components", "smr_seq_fences", "smr_seq_set", "smr_seq_destroy", "smr_debug_kernarg_layout", "smr_debug_canon_prog",
]


class StridedHIPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libstrided_hip error {code}: {msg}")
        self.code = code


class UnsupportedOnDevice(StridedHIPError):
    """SMR_EUNSUPPORTED: valid in the reference but outside the device whitelist."""


def build(jobs: int | None = None, verbose: bool = False) -> str:
    """Compile libstrided_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    jobs = jobs or max(1, os.cpu_count() or 1)
    r = subprocess.run(["make", "-C", CSRC_DIR, f"-j{jobs}"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libstrided_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-8000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


_lib = None


def load():
    """Load the library (once).  Raises loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  strided_jl_amd has no CPU fallback.")
    try:  # share torch's HIP runtime (same SONAME libamdhip64.so.7) when torch is around
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.smr_last_error.restype = C.c_char_p
    lib.smr_abi_version.restype = C.c_int
    lib.smr_init.argtypes = [C.c_int]
    lib.smr_malloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.smr_free.argtypes = [C.c_void_p]
    lib.smr_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.smr_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.smr_stream_sync.argtypes = [C.c_void_p]
    lib.smr_mapreduce.argtypes = [C.POINTER(smr_problem)]
    lib.smr_plan_create.argtypes = [C.POINTER(smr_problem), C.POINTER(C.c_void_p)]
    lib.smr_plan_execute.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
    lib.smr_plan_destroy.argtypes = [C.c_void_p]
    lib.smr_plan_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.smr_plan_algorithmic_bytes.argtypes = [C.c_void_p]
    lib.smr_plan_algorithmic_bytes.restype = C.c_int64
    lib.smr_plan_jit_compile.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.smr_plan_jit_source.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.smr_mapreduce_scalar.argtypes = [C.c_void_p, C.c_void_p]
    lib.smr_plan_tile_order.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t]
    lib.smr_plan_tile_order.restype = C.c_int64
    lib.smr_plan_orbit_pairs.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t]
    lib.smr_plan_orbit_pairs.restype = C.c_int64
    lib.smr_plan_flat_runs.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_size_t]
    lib.smr_plan_flat_runs.restype = C.c_int64
    lib.smr_plan_flat_side.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_size_t]
    lib.smr_plan_flat_side.restype = C.c_int64
    lib.smr_plan_flat_batched.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_size_t]
    lib.smr_plan_flat_batched.restype = C.c_int64
    lib.smr_shard.argtypes = [C.POINTER(smr_problem), C.c_int, C.c_int, C.POINTER(smr_problem),
                              C.POINTER(C.c_int)]
    lib.smr_shard_ex.argtypes = [C.POINTER(smr_problem), C.c_int, C.c_int, C.c_uint32, C.POINTER(smr_problem),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.smr_init_reduction.argtypes = [C.POINTER(smr_problem)]
    lib.smr_mapreduce_sharded.argtypes = [C.POINTER(smr_problem)]
    lib.smr_mapreduce_sharded_ex.argtypes = [C.POINTER(smr_problem), C.c_uint32]
    lib.smr_comm_unique_id.argtypes = [C.c_void_p, C.c_size_t]
    lib.smr_comm_init.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.smr_comm_rank.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.smr_comm_library.argtypes = [C.c_char_p, C.c_size_t]
    lib.smr_set_option.argtypes = [C.c_char_p, C.c_int64]
    lib.smr_get_option.argtypes = [C.c_char_p]
    lib.smr_get_option.restype = C.c_int64
    lib.smr_overlap_begin.argtypes = [C.c_void_p]
    lib.smr_overlap_end.argtypes = [C.c_void_p]
    lib.smr_overlap_fence.argtypes = [C.c_void_p]
    lib.smr_stream_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.smr_stream_destroy.argtypes = [C.c_void_p]
    lib.smr_seq_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.smr_seq_add.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.smr_seq_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.smr_seq_wait.argtypes = [C.c_void_p]
    lib.smr_seq_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.smr_seq_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.smr_seq_components.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_size_t]
    lib.smr_seq_fences.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.smr_seq_destroy.argtypes = [C.c_void_p]
    if lib.smr_abi_version() != 1:
        raise ImportError("libstrided_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc: int):
    if rc == SMR_OK:
        return
    msg = load().smr_last_error().decode("utf-8", "replace")
    if rc == SMR_EUNSUPPORTED:
        raise UnsupportedOnDevice(rc, msg)
    raise StridedHIPError(rc, msg)


def set_option(name: str, value: int):
    check(load().smr_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    return int(load().smr_get_option(name.encode()))


class overlap:
    """`with overlap(stream):` -- an overlap window (smr_overlap_begin / smr_overlap_end): launches issued inside on `stream`
    that touch none of the data of the launches still in flight start without waiting for them (the GPU form of the
    reference's spawn-what-is-independent, src/mapreduce.jl:203-223); results are those of in-order execution.  `stream` is a
    raw hipStream_t handle (None = the null stream; with torch pass torch.cuda.current_stream().cuda_stream).
    On MI355X (gfx950) this is a NO-OP for an ordinary HIP stream: HIP ignores the any-order launch flag on gfx9, so the library skips
    the analysis (option "overlap_window_hip" = 1 re-enables it).  Use `S.Stream()` (the library dispatches itself and overlaps what
    is independent) or a recorded `S.Sequence()`."""

    def __init__(self, stream: int | None = None):
        self.stream = C.c_void_p(stream or 0)

    def __enter__(self):
        check(load().smr_overlap_begin(self.stream))
        return self

    def __exit__(self, *exc):
        check(load().smr_overlap_end(self.stream))
        return False


class Stream:
    """A stream made by the library (smr_stream_create).  On MI355X its launches do not go through HIP: the library submits every one
    itself (AQL packets on HSA queues it owns, the queue chosen by the operands' byte ranges: independent calls overlap, conflicting ones
    are ordered; ~1.8 us of host time per call instead of 3.8) and fences by itself before every copy / synchronisation it performs on it.
    `with S.Stream() as st:` makes the front ends (broadcast `copyto_`, `map_`, `sum`, ...) launch on it; operands that torch produced
    on its own streams must be complete before (torch.cuda.synchronize()), and `st.synchronize()` (done on leaving the block) before
    torch reads the results (`view.toarray()` inside the block synchronises by itself)."""

    def __init__(self):
        h = C.c_void_p()
        check(load().smr_stream_create(C.byref(h)))
        self.handle = int(h.value)

    def synchronize(self):
        check(load().smr_stream_sync(C.c_void_p(self.handle)))

    def __enter__(self):
        import importlib
        importlib.import_module(".mapreduce", __package__)._push_stream(self.handle)  # (the package attribute is the function)
        return self

    def __exit__(self, *exc):
        import importlib
        importlib.import_module(".mapreduce", __package__)._pop_stream()
        self.synchronize()
        return False

    def close(self):
        if self.handle:
            check(load().smr_stream_destroy(C.c_void_p(self.handle)))
            self.handle = 0


class Sequence:
    """smr_seq: a recorded list of plan executions, replayed by the library itself (on MI355X as pre-built AQL packets on its
    own HSA queue; independent launches overlap on the device) with the results of in-order execution on `stream`."""

    def __init__(self):
        self._lib = load()
        self._h = C.c_void_p()
        self._keep = []
        check(self._lib.smr_seq_create(C.byref(self._h)))

    def add(self, plan: "Plan", bases=None):
        arr = None
        if bases is not None:
            arr = (C.c_void_p * len(bases))(*bases)
        check(self._lib.smr_seq_add(self._h, plan._h, arr))
        self._keep.append(plan)
        return self

    def run(self, reps: int = 1, stream: int | None = None):
        check(self._lib.smr_seq_run(self._h, int(reps), C.c_void_p(stream or 0)))

    def wait(self):
        check(self._lib.smr_seq_wait(self._h))

    def info(self) -> str:
        buf = C.create_string_buffer(512)
        check(self._lib.smr_seq_info(self._h, buf, 512))
        return buf.value.decode()

    def set(self, name: str, value: int):
        check(self._lib.smr_seq_set(self._h, name.encode(), int(value)))

    def components(self):
        """Dependency component of every recorded execution (host-only analysis: works without a device)."""
        n = len(self._keep)
        buf = (C.c_int32 * max(1, n))()
        rc = self._lib.smr_seq_components(self._h, buf, n)
        if rc < 0:
            check(rc)
        return list(buf)[:n]

    def fences(self):
        """(acquire flags per recorded execution, footprint in bytes, cache-resident?) -- host-only analysis of what a replay fences."""
        n = len(self._keep)
        buf = (C.c_int32 * max(1, n))()
        fp, res = C.c_int64(0), C.c_int32(0)
        rc = self._lib.smr_seq_fences(self._h, buf, n, C.byref(fp), C.byref(res))
        if rc < 0:
            check(rc)
        return list(buf)[:n], int(fp.value), bool(res.value)

    def __del__(self):
        try:
            if self._h:
                self._lib.smr_seq_destroy(self._h)
                self._h = None
        except Exception:  # pragma: no cover
            pass


class Plan:
    """smr_plan handle: canonicalised problem + chosen kernel, reusable across launches."""

    def __init__(self, problem: smr_problem, keepalive=()):
        self._lib = load()
        self._h = C.c_void_p()
        self._keep = (problem, keepalive)
        check(self._lib.smr_plan_create(C.byref(problem), C.byref(self._h)))

    def execute(self, stream: int | None = None, bases=None):
        arr = None
        if bases is not None:
            arr = (C.c_void_p * len(bases))(*bases)
        check(self._lib.smr_plan_execute(self._h, arr, C.c_void_p(stream or 0)))

    def prepare(self):
        """Upload tables / compile + load the kernel / allocate scratch without launching (call before
        capturing execute() into a hipGraph)."""
        check(self._lib.smr_plan_prepare(self._h))

    def describe(self) -> str:
        buf = C.create_string_buffer(1024)
        check(self._lib.smr_plan_describe(self._h, buf, 1024))
        return buf.value.decode()

    def jit_compile(self) -> int:
        """Compile the plan's kernel for its f-program now (no device needed); returns the size of
        the code object, 0 when the kernel is one of the precompiled ones."""
        n = C.c_size_t(0)
        check(self._lib.smr_plan_jit_compile(self._h, C.byref(n)))
        return int(n.value)

    def jit_source(self) -> str:
        buf = C.create_string_buffer(16384)
        check(self._lib.smr_plan_jit_source(self._h, buf, 16384))
        return buf.value.decode()

    def tile_order(self):
        """Tile executed by each workgroup of a TILED launch (empty list = natural order)."""
        n = int(self._lib.smr_plan_tile_order(self._h, None, 0))
        if n == 0:
            return []
        buf = (C.c_uint32 * n)()
        self._lib.smr_plan_tile_order(self._h, buf, n)
        return list(buf)

    def orbit_pairs(self):
        """ORBIT, PAIR form: eight tiles per workgroup, [w * 8 + b * 4 + g] (empty list: the plan has no PAIR form)."""
        n = int(self._lib.smr_plan_orbit_pairs(self._h, None, 0))
        if n == 0:
            return []
        buf = (C.c_uint32 * n)()
        self._lib.smr_plan_orbit_pairs(self._h, buf, n)
        return list(buf)

    def flat_runs(self):
        """The two-sided FLAT form's runs as a dict (None for any other plan); layout: include/strided_hip.h."""
        n = int(self._lib.smr_plan_flat_runs(self._h, None, 0))
        if n == 0:
            return None
        buf = (C.c_int64 * n)()
        self._lib.smr_plan_flat_runs(self._h, buf, n)
        v = list(buf)
        N = v[2]
        o = 9
        out = dict(kt=v[0], shared=bool(v[1]), N=N, R=v[3:5], TP=v[5:7], p=v[7:9])
        for name in ("dims", "s0", "s1", "in0", "in1"):
            out[name] = v[o:o + N]
            o += N
        out["roff0"] = v[o:o + out["R"][0]]
        o += out["R"][0]
        out["roff1"] = v[o:o + out["R"][1]]
        return out

    def flat_side(self):
        """The one-sided FLAT forms' plan as a dict (None for any other plan); layout: include/strided_hip.h."""
        n = int(self._lib.smr_plan_flat_side(self._h, None, 0))
        if n == 0:
            return None
        buf = (C.c_int64 * n)()
        self._lib.smr_plan_flat_side(self._h, buf, n)
        v = list(buf)
        out = dict(dir=v[0], R=v[1], tplog=v[2], tqlog=v[3], p=v[4], q=v[5], lshare=bool(v[6]), fuse=bool(v[7]), kt=v[8], N=v[9])
        N, o = v[9], 10
        for name in ("dims", "s0", "s1", "ingroup"):
            out[name] = v[o:o + N]
            o += N
        out["roff"] = v[o:o + out["R"]]
        return out

    def flat_batched(self):
        """The batched FLAT form's plan as a dict (None for any other plan); layout: include/strided_hip.h."""
        n = int(self._lib.smr_plan_flat_batched(self._h, None, 0))
        if n == 0:
            return None
        buf = (C.c_int64 * n)()
        self._lib.smr_plan_flat_batched(self._h, buf, n)
        v = list(buf)
        out = dict(g=v[0], P=v[1], K=v[2], N=v[3])
        N, o = v[3], 4
        for name in ("dims", "s0", "s1"):
            out[name] = v[o:o + N]
            o += N
        out["srcoff"] = v[o:o + out["P"]]
        return out

    @property
    def algorithmic_bytes(self) -> int:
        return int(self._lib.smr_plan_algorithmic_bytes(self._h))

    def close(self):
        if self._h:
            self._lib.smr_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
