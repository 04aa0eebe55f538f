"""Sharding on the HIP kernels (reference: the task bisection of _mapreduce_threaded!,
/root/reference/src/mapreduce.jl:195-227, and the per-task partial slots + fold of :153-170).

For N in {2, 4, 8} every sub-problem smr_shard / smr_shard_ex hands out is executed on ONE GPU, one
"rank" after the other, each with its own destination copy; partial destinations are combined with the
product's own kernels and compared with the unsharded oracle.  This is everything of the multi-GPU path
except the wire: the collective itself is one ncclAllReduce (csrc/smr_comm.cpp), exercised here through a
one-rank communicator."""
import numpy as np
import pytest

import oraclelib
import strided_jl_amd as S
from strided_jl_amd import distributed as D
from strided_jl_amd import fn
from util import fview, to_device

pytestmark = pytest.mark.gpu


def _oracle(f, op, initop, dims, arrays):
    p, keep = S.build_problem(f, op, initop, dims, arrays, stream=0)
    oraclelib.mapreduce(p, 1)
    return arrays[0].toarray()


def _dev(arrays):
    cache = {}
    return tuple(to_device(a, cache) for a in arrays)


def _fresh_dest(dest_host):
    """device copy of the destination's parent with the same view"""
    return to_device(dest_host, {})


@pytest.mark.parametrize("nshards", [2, 4, 8])
def test_map_shards_tile_the_destination(nshards):
    import torch
    rng = np.random.default_rng(11)
    A = rng.standard_normal((12, 10, 16, 9))
    A4 = rng.standard_normal((16,) * 4)
    X, Y = rng.standard_normal((80, 24)), rng.standard_normal((40, 24))

    def bcast4_views():
        a = fview(A4)  # ONE buffer, four permuted views (the ORBIT family's case)
        return (fview(np.full((16,) * 4, np.nan)),) + tuple(a.permutedims(p) for p in [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)])

    cases = [
        ("permutedims", lambda x: x, lambda: (fview(np.full((9, 16, 10, 12), np.nan)), fview(A).permutedims((3, 2, 1, 0)))),
        ("bcast4", lambda a, b, c, d: a + b + c + d, bcast4_views),
        ("axpy_strided", lambda x, y: 2.5 * x + y,
         lambda: (fview(np.full((40, 24), np.nan)), fview(X).sview((slice(0, 80, 2), slice(None))), fview(Y))),
    ]
    for name, f, make in cases:
        arrays = make()
        dims = arrays[0].size
        want = _oracle(f, None, None, dims, make())
        dev = _dev(arrays)
        seen = np.zeros(dims, dtype=np.int32)
        for r in range(nshards):
            sdims, sarr, need, sinit = D.shard(f, None, None, dims, dev, nshards, r)
            assert not need and sinit is None
            S._mapreduce_fuse_(f, None, None, sdims, sarr)
            # which destination elements this shard owns
            dim, slabs = D.shard_slices(dims, dev[0].strides, nshards)
            sl = [slice(None)] * len(dims)
            sl[dim] = slice(*slabs[r])
            seen[tuple(sl)] += 1
        torch.cuda.synchronize()
        assert (seen == 1).all(), name  # the slabs tile the box exactly once
        assert np.array_equal(dev[0].toarray(), want), f"{name}: sharded map != oracle"


def _combine(parts, op):
    """fold the per-rank partial destinations with the product's own reduction kernels"""
    stack = np.stack([p for p in parts], axis=-1)
    st = fview(stack)
    cache = {}
    dst = to_device(st, cache)
    out = {"+": S.sum, "*": S.prod, "max": S.maximum, "min": S.minimum}[op](dst, dims=(stack.ndim - 1,))
    import torch
    torch.cuda.synchronize()
    return np.asarray(out.toarray()).reshape(parts[0].shape)


@pytest.mark.parametrize("nshards", [2, 4, 8])
@pytest.mark.parametrize("dtype", [np.float64, np.complex64])
def test_reduction_shards_partials_and_every_initop(nshards, dtype):
    """complete and partial reductions, every initop form of test/othertests.jl:76-102: initop and the
    old destination content enter exactly once (shard 0), the other shards start from the neutral
    element (smr_init_reduction), partials are folded with the reduction op."""
    import torch
    rng = np.random.default_rng(5)
    cx = np.issubdtype(dtype, np.complexfloating)

    def rnd(shape):
        a = rng.standard_normal(shape)
        if cx:
            a = a + 1j * rng.standard_normal(shape)
        return np.asfortranarray(a.astype(dtype))

    X = rnd((24, 18, 16))
    initops = [None, "zero", (lambda x: 1.5 * x), "conj", (lambda x: 0.25)]
    shapes = [
        ("complete", (1, 1, 1), (0, 0, 0)),           # every shard reduces a slab of the slowest dim
        ("keep_dim0", (24, 1, 1), (1, 0, 0)),         # kept dim long enough for 2..8: no all-reduce
        ("keep_dim1_short", (1, 3, 1), None),         # built below: kept extent 3 < nshards for 4, 8
    ]
    for op in ("+", "max"):
        if cx and op == "max":
            continue
        for iname, initop in enumerate(initops):
            if op == "max" and initop is not None and initop != "conj":
                continue
            for sname, kshape, kstr in shapes:
                if sname == "keep_dim1_short":
                    Y = rnd((40, 3, 50))
                    src, dims = Y, Y.shape
                    dest0 = rnd((1, 3, 1))
                    dstr = (0, 1, 0)
                else:
                    src, dims = X, X.shape
                    dest0 = rnd(kshape)
                    dstr = kstr
                f = fn.abs2 if (op == "+" and not cx) else (lambda x: x)
                if op == "max":
                    f = fn.abs
                # unsharded truth (oracle, host)
                hd = fview(dest0.copy())
                hv = S.StridedView(hd.parent, dims, dstr, 0)
                _oracle(f, op, initop, dims, (hv, fview(src)))
                want = hd.toarray()
                # sharded on the device, one rank after the other
                dsrc = to_device(fview(src), {})
                parts = []
                needs = []
                for r in range(nshards):
                    dd = to_device(fview(dest0.copy()), {})
                    dv = S.StridedView(dd.parent, dims, dstr, 0)
                    sdims, sarr, need, sinit = D.shard(f, op, initop, dims, (dv, dsrc), nshards, r)
                    needs.append(need)
                    if need and r != 0:
                        D.init_reduction_(op, dv)
                    S._mapreduce_fuse_(f, op, sinit, sdims, sarr)
                    torch.cuda.synchronize()
                    parts.append(np.asarray(dd.toarray()))
                assert len(set(needs)) == 1
                if needs[0]:
                    got = _combine(parts, op)
                else:
                    # a kept dim was split: shard r owns rows [lo, hi) of its own destination copy
                    got = np.empty_like(parts[0])
                    kd = max(range(len(dims)), key=lambda i: (dstr[i] != 0 and dims[i] >= nshards, abs(dstr[i])))
                    for r in range(nshards):
                        lo, hi = dims[kd] * r // nshards, dims[kd] * (r + 1) // nshards
                        sl = [slice(None)] * got.ndim
                        sl[kd] = slice(lo, hi)
                        got[tuple(sl)] = parts[r][tuple(sl)]
                tol = 1e-5 if np.dtype(dtype).itemsize <= 8 and cx else 1e-12
                want = np.asarray(want).reshape(got.shape)
                assert np.allclose(got, want, rtol=tol, atol=tol * np.abs(want).max()), (op, iname, sname, nshards)


def test_block_partitioned_input_needs_no_replication():
    """smr_shard_ex with a local operand: every "rank" holds only its slab of the input (a separate
    allocation whose index 0 along the split dim is the slab's first box index)."""
    import torch
    rng = np.random.default_rng(3)
    X = np.asfortranarray(rng.standard_normal((32, 20, 16)).astype(np.float32))
    nshards = 4
    total = 0.0
    for r in range(nshards):
        lo, hi = 16 * r // nshards, 16 * (r + 1) // nshards
        slab = torch.from_numpy(np.ascontiguousarray(X[:, :, lo:hi].ravel(order="F"))).cuda()
        out = torch.zeros(1, dtype=torch.float32, device="cuda")
        O = S.StridedView(out, X.shape, (0, 0, 0), 0)
        A = S.StridedView(slab, X.shape, (1, 32, 640), 0)  # logical size = the whole box, memory = the slab
        sdims, sarr, need, sinit = D.shard(fn.abs2, "+", None, X.shape, (O, A), nshards, r, local=(False, True))
        assert need and sdims == (32, 20, hi - lo) and sarr[1].offset == 0
        S._mapreduce_fuse_(fn.abs2, "+", sinit, sdims, sarr)
        torch.cuda.synchronize()
        total += float(out.item())
    ref = float((X.astype(np.float64) ** 2).sum())
    assert abs(total - ref) <= 1e-5 * ref


def test_c_level_collective_path_on_a_one_rank_communicator():
    """smr_comm_init with a real unique id (one rank) + smr_mapreduce_sharded_ex: gather, ncclAllReduce and
    scatter of csrc/smr_comm.cpp run on this GPU (a one-rank communicator still goes through them)."""
    import torch
    rng = np.random.default_rng(9)
    X = np.asfortranarray(rng.standard_normal((64, 48, 8)))
    try:
        uid = D.comm_unique_id()
    except S.UnsupportedOnDevice:
        pytest.skip("librccl.so not available")
    D.comm_init(1, 0, uid)
    try:
        out = torch.full((3,), 7.0, dtype=torch.float64, device="cuda")
        O = S.StridedView(out, X.shape, (0, 0, 0), 1)  # the middle element: a strided (offset) destination
        A = to_device(fview(X), {})
        D.comm_mapreduce_sharded_(fn.abs2, "+", (lambda x: 2 * x), X.shape, (O, A))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert got[0] == 7.0 and got[2] == 7.0
        assert abs(got[1] - (14.0 + (X * X).sum())) <= 1e-10 * (X * X).sum()
        # a map through the same entry point: no collective
        B = to_device(fview(np.zeros((8, 48, 64))), {})
        D.comm_mapreduce_sharded_(lambda x: x, None, None, B.size, (B, A.permutedims((2, 1, 0))))
        torch.cuda.synchronize()
        assert np.array_equal(B.toarray(), X.transpose(2, 1, 0))
    finally:
        D.comm_destroy()
