"""Multi-GPU execution of the funnel: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference's only parallelism is the recursive bisection of the iteration box over tasks
(`_mapreduce_threaded!`, /root/reference/src/mapreduce.jl:195-227): sub-boxes get the same
strides and shifted offsets, never split a reduced dim of a partial reduction (:172-177), and a
complete reduction combines per-task partials afterwards (:153-170).  The same decomposition is
used across ranks (`smr_shard` in the C ABI does the arithmetic):

  * map / permute / broadcast: every rank owns one slab of the destination along its
    slowest-varying dim and computes only that slab -- no collective at all.  Inputs must be
    readable by the rank (replicated, or the rank's own slab when the input is split the same way).
  * reductions: split a kept dim when one is long enough (still no collective); otherwise split a
    reduced dim, every rank reduces its slab into its local destination, and the partial
    destinations are combined with ONE all-reduce (ncclAllReduce, op = the reduction op).
    `initop` and the existing destination content take part exactly once (rank 0).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .mapreduce import _neutral, _redop_code, build_problem, copy_
from .stridedview import StridedView


def _local_mask(local, M):
    if not local:
        return 0
    assert len(local) == M, "`local` needs one flag per operand (destination first)"
    return sum(1 << k for k, f in enumerate(local) if f)


def shard(f, op, initop, dims, arrays, nshards: int, index: int, local=None):
    """Sub-problem `index` of `nshards`: (dims, arrays, needs_allreduce, initop) with the views
    restricted to the shard's sub-box (same parents, shifted offsets).  `local[k]` = operand k is
    block-partitioned: its parent IS this shard's slab (index 0 along the split dim = the slab's first
    box index), so its offset stays put and nothing needs to be replicated."""
    p, keep = build_problem(f, op, initop, dims, arrays, stream=0)
    out = L.smr_problem()
    need = C.c_int(0)
    L.check(L.load().smr_shard_ex(C.byref(p), int(nshards), int(index), _local_mask(local, len(arrays)), C.byref(out),
                                  C.byref(need), None, None, None))
    N = p.N
    sub_dims = tuple(int(out.dims[i]) for i in range(N))
    sub = []
    for k, a in enumerate(arrays):
        strides = a.strides if len(a.strides) == N else (1,) * N
        sub.append(StridedView(a.parent, sub_dims, strides, int(out.ops[k].offset), a.op))
    sub_initop = initop if out.initop != L.SMR_INIT_NONE else None
    return sub_dims, tuple(sub), bool(need.value), sub_initop


def _torch_reduce_op(op):
    import torch.distributed as dist
    return {L.SMR_RED_ADD: dist.ReduceOp.SUM, L.SMR_RED_MUL: dist.ReduceOp.PRODUCT,
            L.SMR_RED_MIN: dist.ReduceOp.MIN, L.SMR_RED_MAX: dist.ReduceOp.MAX,
            L.SMR_RED_AND: dist.ReduceOp.MIN, L.SMR_RED_OR: dist.ReduceOp.MAX}[_redop_code(op)]


def _as_tensor(view: StridedView):
    import torch
    p = view.parent
    return p if isinstance(p, torch.Tensor) else torch.from_numpy(p)


def all_reduce_(view: StridedView, op, group=None) -> StridedView:
    """In-place all-reduce of the elements of a (possibly strided) view: gathered into a dense
    staging buffer, ONE collective, scattered back."""
    import torch.distributed as dist
    from .broadcast import promoteshape1
    dense = view.similar(size=tuple(n for n, s in zip(view.size, view.strides) if s != 0) or (1,))
    # the distinct destination elements: drop the stride-0 (reduced) dims
    kept = [d for d, s in enumerate(view.strides) if s != 0]
    src = StridedView(view.parent, tuple(view.size[d] for d in kept) or (1,),
                      tuple(view.strides[d] for d in kept) or (1,), view.offset, view.op)
    copy_(dense, src)
    t = _as_tensor(dense)
    # no host synchronisation: the gather above runs on torch's current stream, which the collective is
    # ordered after (ProcessGroupNCCL waits for the current stream); the scatter below follows it on the
    # same stream
    dist.all_reduce(t, op=_torch_reduce_op(op), group=group)
    copy_(src, dense)
    return view


def mapreduce_sharded_(f, op, initop, dims, arrays, group=None, local=None):
    """The funnel, executed cooperatively by all ranks of `group`.

    Every rank passes the SAME logical problem over its own device copies of the operands.
    Map: on return each rank's destination holds its own slab (other slabs untouched).
    Reduce: on return every rank's destination holds the complete result.
    """
    import importlib
    import torch.distributed as dist
    from .broadcast import copyto_
    # the package attribute `mapreduce` is the front-end function; fetch the module itself
    mr = importlib.import_module(".mapreduce", __package__)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        mr._mapreduce_fuse_(f, op, initop, dims, arrays)
        return arrays[0]
    sdims, sarrays, need, sinit = shard(f, op, initop, dims, arrays, world, rank, local)
    if need and rank != 0:
        # partial destinations of the other ranks start from the neutral element
        dest = arrays[0]
        kept = [d for d, s in enumerate(dest.strides) if s != 0]
        dview = StridedView(dest.parent, tuple(dest.size[d] for d in kept) or (1,),
                            tuple(dest.strides[d] for d in kept) or (1,), dest.offset, dest.op)
        copyto_(dview, _neutral(op, dest.dtype))
    mr._mapreduce_fuse_(f, op, sinit, sdims, sarrays)
    if need:
        all_reduce_(arrays[0], op, group)
    return arrays[0]


def shard_slices(dims, dest_strides, nshards):
    """Which slab of the destination each rank owns for a MAP (pure host helper for callers that
    want to all-gather the result): returns (dim, [(start, stop), ...])."""
    best, beststride = -1, -1
    for i, (d, s) in enumerate(zip(dims, dest_strides)):
        if s != 0 and d >= nshards and abs(s) > beststride:
            best, beststride = i, abs(s)
    if best < 0:
        best = int(np.argmax(dims))
    d = dims[best]
    return best, [(d * r // nshards, d * (r + 1) // nshards) for r in range(nshards)]


# ---- the C-level path (csrc/smr_comm.cpp): RCCL driven by the library itself -------------------------
def comm_unique_id() -> bytes:
    """128 bytes from rank 0 (ncclGetUniqueId); ship them to the other ranks out of band."""
    buf = C.create_string_buffer(128)
    L.check(L.load().smr_comm_unique_id(buf, 128))
    return bytes(buf.raw)


def comm_init(nranks: int, rank: int, unique_id: bytes | None = None):
    """Every rank, with its device already selected.  `unique_id` may be None only for nranks == 1."""
    if unique_id is None:
        L.check(L.load().smr_comm_init(int(nranks), int(rank), None, 0))
    else:
        buf = C.create_string_buffer(bytes(unique_id), 128)
        L.check(L.load().smr_comm_init(int(nranks), int(rank), buf, 128))


def comm_rank():
    """(rank, nranks) as the library's RCCL communicator reports them (smr_comm_rank)."""
    r, n = C.c_int(-1), C.c_int(0)
    L.check(L.load().smr_comm_rank(C.byref(r), C.byref(n)))
    return int(r.value), int(n.value)


def comm_library() -> str:
    """Path of the RCCL library the C layer uses ($SMR_RCCL_LIB, else an already loaded librccl, else the system's)."""
    buf = C.create_string_buffer(1024)
    L.check(L.load().smr_comm_library(buf, 1024))
    return buf.value.decode()


def comm_destroy():
    L.check(L.load().smr_comm_destroy())


def comm_mapreduce_sharded_(f, op, initop, dims, arrays, local=None, stream=None):
    """smr_mapreduce_sharded_ex: shard -> neutral fill on ranks != 0 -> local kernel -> gather ->
    ncclAllReduce -> scatter, all issued by the library on `stream` (default: torch's current)."""
    p, keep = build_problem(f, op, initop, dims, arrays, stream=stream)
    L.check(L.load().smr_mapreduce_sharded_ex(C.byref(p), _local_mask(local, len(arrays))))
    return arrays[0]


def init_reduction_(op, dest: StridedView, stream=None):
    """Neutral element of `op` into every distinct element of `dest` (smr_init_reduction)."""
    p, keep = build_problem(lambda x: x, op, None, dest.size, (dest, dest), stream=stream)
    L.check(L.load().smr_init_reduction(C.byref(p)))
    return dest
