"""The C-ABI boundary (include/strided_hip.h <-> strided.jl_amd/libstrided_hip.so), CPU-only:
the library loads, exports every declared symbol, validates problems with the documented status
codes, plans without a device, shards with the reference's offset arithmetic, and REFUSES to
compute without a GPU (no CPU fallback)."""
import collections
import ctypes as C
import itertools
import os
import re

import numpy as np
import pytest

import strided_jl_amd as S
from strided_jl_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "strided_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/strided_hip.h but not exported"
    assert sorted(L.EXPORTS) == names
    assert lib.smr_abi_version() == 1


def test_struct_layout_matches_the_header():
    # sizes computed from the header's field list (LP64)
    assert C.sizeof(L.smr_operand) == 8 + 8 + 8 * 8 + 4 + 4
    assert C.sizeof(L.smr_problem) == 4 + 4 + 8 * 8 + 8 * C.sizeof(L.smr_operand) + 8 + 4 + 4 + 8 + 4 + 4 + 16 + 8


def _views(dims, strides, dtype=np.float64):
    out = []
    for st in strides:
        span = 1 + sum((d - 1) * abs(s) for d, s in zip(dims, st))
        out.append(S.StridedView(np.zeros(span, dtype=dtype), dims, st, 0))
    return tuple(out)


def test_planning_needs_no_device_and_picks_the_expected_family():
    x = S.StridedView(np.zeros((32, 32, 32, 32), order="F"))
    y = x.similar()
    d = S.make_plan(lambda v: v, None, None, x.size, (y, x.permutedims((3, 2, 1, 0)))).describe()
    assert "family=tiled" in d and "f=ident" in d and "tile=d0:32,d3:32" in d and "algbytes=16777216" in d
    ps = [x.permutedims(q) for q in [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]]
    d = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, x.size, (y, *ps)).describe()
    assert "family=orbit" in d and "f=add4" in d and "group=4" in d and "tile=d0:4,d1:4,d2:4,d3:4" in d and "algbytes=16777216" in d
    S.set_option("orbit", 0)  # the classic tiled kernel stays available (distinct buffers, ragged sizes, tuning)
    try:
        d = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, x.size, (y, *ps)).describe()
        assert "family=tiled" in d and "f=add4" in d and "staged=3" in d and "algbytes=16777216" in d
    finally:
        S.set_option("orbit", 1)
    # four DISTINCT arrays with the same permuted strides: no shared buffer, classic kernel
    zs = [x.similar().permutedims(q) for q in [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]]
    d = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, x.size, (y, *zs)).describe()
    assert "family=tiled" in d and "staged=3" in d
    d = S.make_plan(lambda a: a * S.fn.exp(-2 * a) + S.fn.sin(a * a), None, None, x.size, (y, x)).describe()
    assert "family=stream" in d and "f=expr5" in d and "N=1" in d  # 4 dims fuse into one
    o = x.similar(size=(1,))
    d = S.make_plan(S.fn.abs2, "+", None, x.size, S.promoteshape(x.size, o.sreshape((1, 1, 1, 1)), x)).describe()
    assert "family=reduce_all" in d and "f=abs2" in d
    # the four appearances of A in the README's compute-bound expression are deduplicated
    d = S.make_plan(lambda a, b, c, e: a * S.fn.exp(-2 * b) + S.fn.sin(c * e), None, None, x.size, (y, x, x, x, x)).describe()
    assert "M=2" in d and "f=expr5" in d


def _sym_plan(m, dtype=np.float64):
    a = S.StridedView(np.zeros((m, m), dtype=dtype, order="F"))
    b = a.similar()
    return S.make_plan(lambda x, y: (x + y) / 2, None, None, a.size, (b, a, a.permutedims((1, 0))))


@pytest.fixture
def classic_tiled():
    S.set_option("orbit", 0)
    yield
    S.set_option("orbit", 1)


@pytest.mark.parametrize("m", [200, 1000, 4000])
def test_tile_order_is_a_permutation_that_keeps_transposed_partners_on_one_xcd(m, classic_tiled):
    """B .= (A .+ A')./2: tile (i, j) and tile (j, i) read the same two regions of A; the planner
    must run them back to back on one XCD (workgroup b -> XCD b mod 8) and every tile once."""
    plan = _sym_plan(m)
    d = plan.describe()
    assert "order=orbits:" in d
    nt = -(-m // 32)
    assert f"grid={nt * nt} " in d
    ord_ = np.array(plan.tile_order(), dtype=np.int64)
    assert len(ord_) % 8 == 0 and len(ord_) - nt * nt < 8
    real = ord_[ord_ != 0xFFFFFFFF]
    assert sorted(real.tolist()) == list(range(nt * nt))          # every tile exactly once
    assert int(plan.describe().split("order=orbits:")[1].split()[0]) == nt * (nt + 1) // 2
    where = np.full(nt * nt, -1, dtype=np.int64)
    where[real] = np.nonzero(ord_ != 0xFFFFFFFF)[0]
    i, j = np.meshgrid(np.arange(nt), np.arange(nt), indexing="ij")
    t, tt = (i + nt * j).ravel(), (j + nt * i).ravel()
    off = t != tt
    same_xcd = (where[t[off]] % 8) == (where[tt[off]] % 8)
    adjacent = np.abs(where[t[off]] // 8 - where[tt[off]] // 8) == 1
    # only pairs cut by one of the 7 run boundaries may be separated
    assert (~(same_xcd & adjacent)).sum() <= 2 * 7


def _orbit_cover(roots, nt, perms):
    """tiles covered by the orbits of `roots` (linear ids, dim 0 fastest) under the dim permutations"""
    nd = len(nt)
    mul = np.cumprod([1] + list(nt[:-1]))
    covered = []
    for r in roots:
        t = [(r // mul[d]) % nt[d] for d in range(nd)]
        orb = set()
        for g in perms:
            u = [0] * nd
            for d in range(nd):
                u[g[d]] = t[d]
            orb.add(int(sum(u[d] * mul[d] for d in range(nd))))
        covered.extend(sorted(orb))
    return covered


def test_orbit_list_covers_every_tile_once_and_groups_line_partners():
    """FAM_ORBIT: a workgroup holds one tile per LDS slot -- the orbit of a tile under the group the permuted views generate;
    orbits on a diagonal (fewer distinct tiles than slots) share workgroups (round 6), so every tile of the box is loaded and
    stored exactly once, the 32^4 launch is 1024 workgroups (4 per CU) and the orbits of one super-cell (2 tiles along every
    tiled dim) run on one XCD."""
    x = S.StridedView(np.zeros((32, 32, 32, 32), order="F"))
    y = x.similar()
    ps = [x.permutedims(q) for q in [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]]
    plan = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, x.size, (y, *ps))
    # necklaces of length 4 over 8 symbols: 1008 of period 4, 28 of period 2, 8 of period 1 -> 1008 + 14 + 2 workgroups
    assert "family=orbit" in plan.describe() and "orbits=1044" in plan.describe() and "grid=1024" in plan.describe()
    lst = np.array(plan.tile_order(), dtype=np.int64).reshape(-1, 4)
    assert len(lst) == 1024 and len(lst) % 8 == 0
    assert sorted(lst.ravel().tolist()) == list(range(8 ** 4))  # every tile once, no idle workgroup
    # a workgroup's tiles are closed under the rotation of the four coordinates (whole orbits only)
    def rot(t):
        return (t % 8) * 512 + t // 8   # (t0,t1,t2,t3) -> (t1,t2,t3,t0)
    for row in lst[::37]:
        assert {rot(int(t)) for t in row} == {int(t) for t in row}
    # super-cell (2^4 tiles) of a tile; a cell's tiles occupy workgroups of ONE XCD (the eight runs may cut 7 cells, and a
    # shared workgroup holds diagonal tiles of a few neighbouring cells)
    cells = {}
    for b, row in enumerate(lst):
        for t in row:
            tc = [(int(t) // 8 ** d) % 8 for d in range(4)]
            cells.setdefault(tuple(c // 2 for c in tc), set()).add(b % 8)
    split = sum(1 for v in cells.values() if len(v) > 1)
    assert split <= 7 + 16, split
    S.set_option("orbit_pack", 0)
    try:
        plan = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, x.size, (y, *ps))
        lst = np.array(plan.tile_order(), dtype=np.int64).reshape(-1, 4)
        assert "grid=1048" in plan.describe() and len(lst) == 1048
        live = lst[lst[:, 0] != 0xFFFFFFFF]
        assert len(live) == 1044 and sorted(set(live.ravel().tolist())) == list(range(8 ** 4))
    finally:
        S.set_option("orbit_pack", 1)
    # symmetrise: pairs of transposed 32x32 tiles; the 125 diagonal tiles share 63 workgroups
    plan = _sym_plan(4000)
    assert "family=orbit" in plan.describe() and "group=2" in plan.describe() and "tile=d0:32,d1:32" in plan.describe()
    lst = np.array(plan.tile_order(), dtype=np.int64).reshape(-1, 2)
    live = lst[lst[:, 0] != 0xFFFFFFFF]
    assert len(live) == 125 * 124 // 2 + 63
    tiles = live.ravel().tolist()
    assert sorted(set(tiles)) == list(range(125 * 125)) and len(tiles) == 125 * 125 + 1  # the odd diagonal tile repeats in its workgroup
    # sizes without a power-of-two divisor fall back to the classic kernel with ragged tiles
    assert "family=tiled" in _sym_plan(1000).describe()


def test_tile_order_four_way_permuted_sum_and_opt_out(classic_tiled):
    x = S.StridedView(np.zeros((32, 32, 32, 32), order="F"))
    y = x.similar()
    ps = [x.permutedims(q) for q in [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]]
    plan = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, x.size, (y, *ps))
    assert "order=orbits:6" in plan.describe()  # necklaces of {0,1}^4: 16^4 super-tiles of a 32^4 box
    ord_ = plan.tile_order()
    assert sorted(ord_) == list(range(256))
    # a plain permutedims! has one input: nothing to co-locate, natural order
    plan = S.make_plan(lambda v: v, None, None, x.size, (y, x.permutedims((3, 2, 1, 0))))
    assert plan.tile_order() == [] and "order=" not in plan.describe()
    # distinct buffers with permuted strides are not aliases
    z = x.similar()
    plan = S.make_plan(lambda a, b: a + b, None, None, x.size, (y, x.permutedims((1, 2, 3, 0)), z.permutedims((2, 3, 0, 1))))
    assert plan.tile_order() == []
    S.set_option("tile_order", 0)
    try:
        assert _sym_plan(1000).tile_order() == []
    finally:
        S.set_option("tile_order", 1)


def test_block_tile_order_for_distinct_arrays_with_several_unit_axes():
    """Round 3: inputs that are DISTINCT arrays with different unit axes have no orbit structure; their
    tiles are visited in compact blocks (4 tiles per dim) so that tiles in flight together complete each other's short
    runs.  Every tile exactly once; a block's tiles are adjacent in the list; round-robin over the XCDs once the
    operands exceed the Infinity Cache, one contiguous run per XCD below it; opt-out with tile_block=0."""
    n = 64
    xs = [S.StridedView(np.zeros((n,) * 4, order="F")) for _ in range(4)]
    y = xs[0].similar()
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    plan = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, y.size, (y, *[x.permutedims(q) for x, q in zip(xs, perms)]))
    d = plan.describe()
    assert "family=tiled" in d and "tile=d0:16,d1:8,d2:8,d3:4" in d and "order=orbits:" in d    # big tiles + block order
    ord_ = np.array(plan.tile_order(), dtype=np.int64)
    nt = (4, 8, 8, 16)
    assert sorted(ord_[ord_ != 0xFFFFFFFF]) == list(range(4 * 8 * 8 * 16))
    first = ord_[:256]                                        # one block = 4 x 4 x 4 x 4 tiles
    co = np.stack(np.unravel_index(first, nt, order="F"))
    assert all(co[k].max() - co[k].min() <= 3 for k in range(4))
    S.set_option("tile_block", 0)
    try:
        assert S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, y.size, (y, *[x.permutedims(q) for x, q in zip(xs, perms)])).tile_order() == []
    finally:
        S.set_option("tile_block", -1)
    # cache-resident operands (5 x 21 MiB): one contiguous run of the list per XCD (workgroup b -> XCD b mod 8)
    m = 40
    xs = [S.StridedView(np.zeros((m,) * 4, order="F")) for _ in range(4)]
    y = xs[0].similar()
    plan = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, y.size, (y, *[x.permutedims(q) for x, q in zip(xs, perms)]))
    ord_ = np.array(plan.tile_order(), dtype=np.int64)
    real = ord_[ord_ != 0xFFFFFFFF]
    assert len(real) == len(set(real.tolist())) and "order=orbits:" in plan.describe()
    run0 = ord_[0::8][:16]                                    # what XCD 0 executes first: consecutive members of one block
    nt = tuple(int(v) for v in plan.describe().split("grid=")[0].split("tile=")[1].replace("d0:", "").replace("d1:", "").replace("d2:", "").replace("d3:", "").split()[0].split(","))
    tiles = tuple(-(-m // e) for e in nt)
    co = np.stack(np.unravel_index(run0, tiles, order="F"))
    assert all(co[k].max() - co[k].min() <= 3 for k in range(4))


def test_invalid_problems_return_einval():
    x, y = _views((8, 8), [(1, 8), (1, 8)])
    p, keep = S.build_problem(lambda v: v, None, None, (8, 8), (x, y), stream=0)
    h = C.c_void_p()
    lib = L.load()
    p.N = 0
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == L.SMR_EINVAL
    p.N = 2
    p.dims[1] = 0
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == L.SMR_EINVAL
    assert b">= 1" in lib.smr_last_error()
    p.dims[1] = 8
    p.ops[1].dtype = 99
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == L.SMR_EINVAL
    p.ops[1].dtype = L.SMR_F64
    p.initop = L.SMR_INIT_ZERO  # initop without a reduction
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == L.SMR_EINVAL
    p.initop = 0
    bad = (C.c_uint8 * 2)(L.OPCODES["ADD"], 0)  # stack underflow
    p.fprog = C.cast(bad, C.POINTER(C.c_uint8))
    p.fprog_len = 1
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == L.SMR_EINVAL
    assert lib.smr_set_option(b"no_such_option", 1) == L.SMR_EINVAL


def test_map_into_zero_stride_destination_is_unsupported():
    x, y = _views((8, 8), [(1, 0), (1, 8)])
    with pytest.raises(L.UnsupportedOnDevice):
        S.make_plan(lambda v: v, None, None, (8, 8), (x, y))


def test_compute_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    x, y = _views((8, 8), [(1, 8), (8, 1)])
    with pytest.raises(RuntimeError, match="MI355X only"):
        S.copy_(x, y)  # host views never reach the engine
    plan = S.make_plan(lambda v: v, None, None, (8, 8), (x, y))
    with pytest.raises(L.StridedHIPError) as e:
        plan.execute()
    assert e.value.code == L.SMR_ENODEVICE


def test_shard_follows_the_reference_offset_arithmetic():
    from strided_jl_amd import distributed as D
    dims = (4096, 4096, 64)
    A, = _views(dims, [(1, 4096, 4096 * 4096)], np.float32)
    out = S.StridedView(np.zeros(1, np.float32), dims, (0, 0, 0), 0)
    total = 0
    for r in range(8):
        sdims, (so, sa), need, sinit = D.shard(S.fn.abs2, "+", "zero", dims, (out, A), 8, r)
        assert need  # complete reduction: a reduced dim is split -> all-reduce of the partial
        assert sdims == (4096, 4096, 8)
        assert sa.offset == r * 8 * 4096 * 4096 and so.offset == 0
        assert (sinit == "zero") == (r == 0)  # initop exactly once
        total += sdims[2]
    assert total == 64
    # map: split the slowest destination dim, offsets += start * stride for every operand
    x, y = _views((100, 30, 7), [(1, 100, 3000), (210, 7, 1)])
    covered = 0
    for r in range(3):
        sdims, (sx, sy), need, _ = D.shard(lambda v: v, None, None, (100, 30, 7), (x, y), 3, r)
        assert not need and sdims[:2] == (100, 30)
        start = 7 * r // 3
        assert sx.offset == start * 3000 and sy.offset == start * 1
        covered += sdims[2]
    assert covered == 7
    # partial reduction with a long kept dim: split that one, no collective
    o, a = _views((64, 50), [(1, 0), (1, 64)])
    sdims, _, need, _ = D.shard(lambda v: v, "+", None, (64, 50), (o, a), 4, 1)
    assert not need and sdims == (16, 50)


def test_comm_entry_points_without_a_communicator():
    """The C-level multi-GPU entry points (csrc/smr_comm.cpp): argument checking and the
    single-rank degenerate case need neither RCCL nor a device."""
    lib = L.load()
    rank, n = C.c_int(-1), C.c_int(-1)
    assert lib.smr_comm_rank(C.byref(rank), C.byref(n)) == L.SMR_OK and (rank.value, n.value) == (0, 1)
    assert lib.smr_comm_init(0, 0, None, 0) == L.SMR_EINVAL
    assert lib.smr_comm_init(2, 2, None, 0) == L.SMR_EINVAL
    assert lib.smr_comm_init(2, 0, None, 0) == L.SMR_EINVAL          # more than one rank needs the unique id
    assert lib.smr_comm_init(1, 0, None, 0) == L.SMR_OK              # one rank: nothing to set up
    assert lib.smr_comm_destroy() == L.SMR_OK
    assert lib.smr_comm_unique_id(None, 0) == L.SMR_EINVAL


def test_planner_picks_the_vectorised_reduction_and_stream_forms():
    """Planning is host arithmetic: the forms added for views (ROW / COL partial reductions, complete
    reductions over sub-boxes, strided and short rows in the STREAM family) are chosen when expected."""
    a = S.StridedView(np.zeros((512, 384), dtype=np.float32, order="F"))

    def red(view, dims):
        out = view.similar(size=tuple(1 if d in dims else n for d, n in enumerate(view.size)))
        return S.make_plan(lambda x: x, "+", None, view.size, S.promoteshape(view.size, out, view)).describe()

    assert "family=reduce_part" in red(a, (0,)) and "form=row" in red(a, (0,))      # sum(A; dims=1): contiguous reduced dim
    assert "form=col" in red(a, (1,))                                                # sum(A; dims=2): contiguous kept dim
    assert "form=general" in red(a.sview(slice(0, 512, 2), slice(None)), (0,))       # stride 2 along the reduced dim
    assert "family=reduce_all" in red(a, (0, 1)) and "N=1" in red(a, (0, 1))         # fuses into one run
    box = a.sview(slice(0, 500), slice(0, 300))
    d = red(box, (0, 1))                                                             # sub-box: does not fuse
    assert "family=reduce_part" in d and "nout=1" in d and "form=row" in d
    # few outputs, long reductions: split into many workgroups + folding pass
    t = S.StridedView(np.zeros((16, 1 << 16), dtype=np.float32, order="F"))
    assert "split=" in red(t, (1,)) and "split=1 " not in red(t, (1,)) + " "
    # STREAM: stepped rows and short rows stay in the family
    b = a.similar()
    d = S.make_plan(lambda x: x * 2, None, None, (256, 384), (b.sview(slice(0, 256), slice(None)), a.sview(slice(0, 512, 2), slice(None)))).describe()
    assert "family=stream" in d and "vec=1" in d
    d = S.make_plan(lambda x: x * 2, None, None, (100, 384), (b.sview(slice(0, 100), slice(None)), a.sview(slice(0, 100), slice(None)))).describe()
    assert "family=stream" in d and "vec=4" in d


def test_shard_ex_local_operands_and_reported_slab():
    """smr_shard_ex: flagged operands keep their offset (block-partitioned inputs), the others get the
    reference's shift (src/mapreduce.jl:217-219); the slab [start, stop) is reported."""
    x, y = _views((6, 5, 16), [(0, 0, 0), (1, 6, 30)], np.float32)
    p, keep = S.build_problem(S.fn.abs2, "+", None, (6, 5, 16), (x, y), stream=0)
    lib = L.load()
    for nsh in (2, 4, 8):
        for r in range(nsh):
            out = L.smr_problem()
            need, dim, lo, hi = C.c_int(0), C.c_int(-2), C.c_int64(-1), C.c_int64(-1)
            assert lib.smr_shard_ex(C.byref(p), nsh, r, 0b10, C.byref(out), C.byref(need), C.byref(dim), C.byref(lo), C.byref(hi)) == 0
            assert need.value == 1 and dim.value == 2
            assert (lo.value, hi.value) == (16 * r // nsh, 16 * (r + 1) // nsh)
            assert out.dims[2] == hi.value - lo.value and out.ops[1].offset == 0 and out.ops[0].offset == 0
            out2 = L.smr_problem()
            assert lib.smr_shard(C.byref(p), nsh, r, C.byref(out2), C.byref(need)) == 0
            assert out2.ops[1].offset == 30 * lo.value  # replicated operand: shifted inside the whole parent


def test_64bit_integer_inputs_compute_in_the_integer_class_or_are_refused():
    """Round 3: integer operands + integer-closed f run in the wrapping Int64 class (tests/test_integer_class.py); a 64-bit
    integer input that meets floating-point arithmetic would run in Float64 (exact only below 2^53) and stays
    SMR_EUNSUPPORTED (the Julia shim falls back to the CPU method)."""
    lib = L.load()
    h = C.c_void_p()
    a, b = _views((16, 16), [(1, 16), (16, 1)], np.int64)
    p, keep = S.build_problem(lambda v: v, None, None, (16, 16), (a, b), stream=0)     # permutedims! of Int64: a bit copy
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == 0
    lib.smr_plan_destroy(h)
    p, keep = S.build_problem(lambda v: v + 1, None, None, (16, 16), (a, b), stream=0)   # integer class
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == 0
    lib.smr_plan_destroy(h)
    p, keep = S.build_problem(lambda v: v / 2, None, None, (16, 16), (a, b), stream=0)   # `/` leaves the integers
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == L.SMR_EUNSUPPORTED and b"2^53" in lib.smr_last_error()
    f, = _views((16, 16), [(1, 16)], np.float64)
    p, keep = S.build_problem(lambda v, w: v + w, None, None, (16, 16), (f, b, f), stream=0)   # Int64 + Float64
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == L.SMR_EUNSUPPORTED
    o = S.StridedView(np.zeros(1, dtype=np.int64), (16, 16), (0, 0), 0)
    p, keep = S.build_problem(lambda v: v, "+", None, (16, 16), (o, b), stream=0)       # sum of Int64: integer class
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == 0
    lib.smr_plan_destroy(h)
    c32, = _views((16, 16), [(1, 16)], np.int32)
    p, keep = S.build_problem(lambda v: v, "+", None, (16, 16), (o, c32), stream=0)     # Int32 -> Int64 sum
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == 0
    lib.smr_plan_destroy(h)


def test_plan_rebinding_must_keep_the_aliasing_pattern():
    """canonicalise merges identical views of one buffer (A .+ A); rebinding such a plan to two different
    buffers would silently ignore one of them -> SMR_EINVAL."""
    lib = L.load()
    x, y = _views((64, 64), [(1, 64), (1, 64)])
    p, keep = S.build_problem(lambda u, v: u + v, None, None, (64, 64), (x, y, y), stream=0)
    h = C.c_void_p()
    assert lib.smr_plan_create(C.byref(p), C.byref(h)) == 0
    buf = (C.c_char * 64)()
    addr = C.addressof(buf)
    same = (C.c_void_p * 3)(addr, addr + 8, addr + 8)
    diff = (C.c_void_p * 3)(addr, addr + 8, addr + 16)
    # (no device here: the aliasing check comes before anything touches the GPU; a consistent rebinding gets
    # as far as the device check)
    assert lib.smr_plan_execute(h, diff, None) == L.SMR_EINVAL and b"share a buffer" in lib.smr_last_error()
    assert lib.smr_plan_execute(h, same, None) != L.SMR_EINVAL or b"share a buffer" not in lib.smr_last_error()
    lib.smr_plan_destroy(h)


def test_planner_rules_for_short_dims_big_transposes_and_short_reductions():
    """Round-2 planner rules found by tools/perf_sanity.py (host arithmetic only)."""
    def perm(dims, p, dt=np.float64):
        a = S.StridedView(np.zeros(dims, dtype=dt, order="F"))
        b = S.StridedView(np.zeros(tuple(dims[i] for i in p), dtype=dt, order="F"))
        return S.make_plan(lambda x: x, None, None, b.size, (b, a.permutedims(p))).describe()

    # a short common unit axis with different continuation dims behind it is a transposition one level up
    d = perm((2, 128, 2, 128, 8), (0, 3, 1, 2, 4))
    assert "family=tiled" in d and "tile=d0:2,d1:32,d2:16" in d, d
    assert "family=stream" in perm((8, 64, 64), (0, 2, 1))            # 64-byte rows: STREAM keeps them
    assert "family=stream" in perm((100, 90, 80), (0, 2, 1))
    # a nearly idle short-row STREAM workgroup goes elsewhere
    assert "family=stream" not in perm((4,) * 8, (0, 3, 5, 4, 2, 6, 7, 1))
    # a unit axis shorter than the run target continues into the dim whose stride equals its extent
    S.set_option("flat", 0)
    try:
        assert "tile=d0:32,d1:8,d2:4" in perm((3, 1000, 700), (2, 1, 0))
    finally:
        S.set_option("flat", 1)
    # round 3: short leading dims with extents that are not powers of two are addressed as ONE flat run (FLAT family)
    d = perm((3, 1000, 700), (2, 1, 0))
    assert "family=flat" in d and "flat_side=input run=3x16(d1) line=d0:32" in d, d
    d = perm((640, 480, 3), (2, 1, 0))
    assert "family=flat" in d and "flat_side=dest run=3x16(d1) line=d2:32" in d, d
    d = perm((100, 3, 100, 3, 10), (4, 3, 2, 1, 0))
    assert "family=flat" in d and "run=30x2(d2)" in d, d
    assert "family=flat" not in perm((4, 1000, 700), (2, 1, 0))      # powers of two stay with the tiled family
    # ... and when BOTH sides' unit-stride dims are short, each side gets a run of its own (two-sided form)
    d = perm((5, 300, 300, 7), (3, 2, 1, 0))
    assert "family=flat" in d and "two-sided dest_run=7x6(d1) input_run=5x9(d2)" in d, d
    d = perm((17, 33, 65, 31), (3, 2, 1, 0))
    assert "two-sided dest_run=31x1(d1) input_run=17x2(d2)" in d, d
    assert "two-sided" in perm((6, 64, 64, 64, 5), (4, 3, 2, 1, 0), np.complex128)   # one-sided line of 5 elements: two-sided first
    assert "two-sided" not in perm((24, 100, 100, 20), (3, 2, 1, 0), np.float32)     # 24-element lines: one-sided with 16-byte accesses
    S.set_option("flat2", 0)
    try:
        assert "family=tiled" in perm((5, 300, 300, 7), (3, 2, 1, 0))
    finally:
        S.set_option("flat2", 1)
    # n-ary maps: ONE input with the other layout, the rest laid out like the destination (tensoradd!: C .= beta .* C .+ alpha .* permutedims(A, p))
    cA, cC, cC2 = (S.StridedView(np.zeros(sh, dtype=np.float64, order="F")) for sh in ((640, 480, 3), (3, 480, 640), (3, 480, 640)))
    d = S.make_plan(lambda c, a: 0.5 * c + 2.0 * a, None, None, cC.size, (cC, cC, cA.permutedims((2, 1, 0)))).describe()
    assert "family=flat" in d and "f=axpby" in d and "M=3" in d, d
    d = S.make_plan(lambda c, a, e: c + a * e, None, None, cC.size, (cC, cC, cA.permutedims((2, 1, 0)), cC2)).describe()
    assert "family=flat" in d and "M=4" in d, d
    tA = S.StridedView(np.zeros((5, 60, 50, 7), dtype=np.float64, order="F"))
    tC = S.StridedView(np.zeros((7, 50, 60, 5), dtype=np.float64, order="F"))
    d = S.make_plan(lambda c, a: c + a, None, None, tC.size, (tC, tC, tA.permutedims((3, 2, 1, 0)))).describe()
    assert "two-sided" in d and "M=3" in d, d
    # two inputs with layouts of their own: not this family
    d = S.make_plan(lambda a, b: a + b, None, None, cC.size, (cC, cA.permutedims((2, 1, 0)), S.StridedView(np.zeros((480, 3, 640), order="F")).permutedims((1, 0, 2)))).describe()
    assert "family=flat" not in d, d
    # HBM-sized transposes of 8-/16-byte elements: 128 x 32 tiles on 1024 lanes; Float32 and smaller problems: 32 x 32
    assert "tile=d0:128,d1:32" in perm((8192, 8192), (1, 0)) and "threads=1024" in perm((8192, 8192), (1, 0))
    assert "tile=d0:32,d1:32" in perm((8192, 8192), (1, 0), np.float32)
    assert "tile=d0:32,d1:32" in perm((4000, 4000), (1, 0))

    def red(dims, rd):
        a = S.StridedView(np.zeros(dims, dtype=np.float32, order="F"))
        out = a.similar(size=tuple(1 if i in rd else n for i, n in enumerate(dims)))
        return S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, a)).describe()

    assert "form=col lanes_per_out=1 " in red((100, 90, 80, 7), (3,))           # 7 rows per output: one lane walks them
    assert "form=col lanes_per_out=8 " in red((512, 384, 64), (2,))             # 64 rows: shared through LDS as before
    assert "split=7 " in red((100, 90, 80, 7), (1, 3)) + " "                      # outer (7) x inner (1) cuts together
    # round 3 (tools/reduce_sweep.py): no floor of 16 lanes per output in the ROW form ...
    assert "form=row lanes_per_out=1 split=1" in red((3, 1920, 1080), (0,))      # sum over 3 channels: one lane per pixel (200 -> 15 us)
    assert "form=row lanes_per_out=4 split=1" in red((100, 90, 80, 7), (0,))     # 100 floats: 4 lanes x 6 vectors (7.6 -> 4.75 us)
    assert "form=row lanes_per_out=2 split=1" in red((32, 200000), (0,))
    # ... narrower COL row segments when that fills the device without a split ...
    assert "form=col lanes_per_out=32 split=1" in red((512, 384, 64), (1,))      # 16.0 -> 12.4 us
    assert "form=col lanes_per_out=16 split=1" in red((256, 256, 256), (1,))     # 18.4 -> 14.2 us
    # ... and splits aim at 1024 (ROW) / 512 (COL) workgroups instead of 4096
    # (round 6: rows of 100 Float32 = 25 vectors get 25 lanes x 10 rows instead of 32 x 8)
    assert "form=col lanes_per_out=10 split=512 lanes=25x10" in red((100, 90, 80, 7), (1, 2, 3))
    assert "form=row lanes_per_out=256 split=2 " in red((512, 384, 64), (0, 2)) + " "
    # at most `reduce_single` chunks: folded inside the launch; the option round-trips and 0 restores the two-launch form
    assert S.get_option("reduce_single") == 4
    S.set_option("reduce_single", 0)
    assert S.get_option("reduce_single") == 0
    S.set_option("reduce_single", 4)


def test_plans_of_the_baseline_configs():
    """The five BASELINE.json configs land in the kernel family / tile the measurements were taken with."""
    fn = S.fn

    def cm(dims, dt=np.float64):
        return S.StridedView(np.zeros(dims, dtype=dt, order="F"))

    n = 32
    A, B = cm((n,) * 4), cm((n,) * 4)
    d = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0)))).describe()            # configs[1]
    assert "family=tiled" in d and "tile=d0:32,d3:32" in d and "grid=1024" in d, d
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    d = S.make_plan(lambda a, b, c, e: a + b + c + e, None, None, A.size, (B,) + tuple(A.permutedims(p) for p in perms)).describe()  # configs[2]
    assert "family=orbit" in d and "f=add4" in d and "tile=d0:4,d1:4,d2:4,d3:4" in d and "orbits=1044" in d, d
    M, N = cm((4000, 4000)), cm((4000, 4000))
    d = S.make_plan(lambda x, y: (x + y) / 2, None, None, (4000, 4000), (N, M, M.adjoint())).describe()        # configs[0]
    assert "family=orbit" in d and "f=sym" in d and "group=2" in d, d
    X = S.StridedView(np.zeros(1 << 20, dtype=np.float32))
    O = S.StridedView(np.zeros(1, dtype=np.float32), (1 << 20,), (0,), 0)
    d = S.make_plan(fn.abs2, "+", None, (1 << 20,), (O, X)).describe()                                         # configs[3], one shard
    assert "family=reduce_all" in d and "f=abs2" in d, d
    E, F_ = cm((2048, 2048), np.float32), cm((2048, 2048), np.float32)
    d = S.make_plan(lambda a: a * fn.exp(-2 * a) + fn.sin(a * a), None, None, (2048, 2048), (F_, E)).describe()  # configs[4]
    assert "family=stream" in d and "f=expr5" in d and "vec=4" in d and "N=1" in d, d


def _walk_two_sided_flat(fr):
    """CPU model of smr_k_flat.hip's flat2_body + its launcher: every (destination offset, input offset) pair the launch touches, tile
    by tile, from the numbers smr_plan_flat_runs reports."""
    dims, s0, s1 = fr["dims"], fr["s0"], fr["s1"]
    R, TP, p = fr["R"], fr["TP"], fr["p"]
    outer = [d for d in range(fr["N"]) if not fr["in0"][d] and not fr["in1"][d] and d not in (p[0], p[1])]
    dimp = [dims[p[t]] if p[t] >= 0 else 1 for t in (0, 1)]
    spo = [(s1 if t == 0 else s0)[p[t]] if p[t] >= 0 else 0 for t in (0, 1)]     # stride of p[t] on the OTHER side
    ntp = [(dimp[t] + TP[t] - 1) // TP[t] for t in (0, 1)]
    roff0, roff1 = np.array(fr["roff0"], dtype=np.int64), np.array(fr["roff1"], dtype=np.int64)
    dst, src = [], []
    for oidx in itertools.product(*[range(dims[d]) for d in outer]):
        bd0 = sum(i * s0[d] for i, d in zip(oidx, outer))
        bs0 = sum(i * s1[d] for i, d in zip(oidx, outer))
        for t0 in range(ntp[0]):
            for t1 in range(ntp[1]):
                p0, p1 = t0 * TP[0], t1 * TP[1]
                bd = bd0 + p0 * R[0] + p1 * spo[1]
                bs = bs0 + p1 * R[1] + p0 * spo[0]
                n0 = min(TP[0], dimp[0] - p0) * R[0]
                n1 = min(TP[1], dimp[1] - p1) * R[1]
                x = np.arange(n0, dtype=np.int64)
                y = np.arange(n1, dtype=np.int64)
                offs = roff0[x % R[0]] + (x // R[0]) * spo[0]       # input offset of position x of the destination run
                offd = roff1[y % R[1]] + (y // R[1]) * spo[1]       # destination offset of position y of the input run
                dst.append((bd + x[:, None] + offd[None, :]).ravel())
                src.append((bs + y[None, :] + offs[:, None]).ravel())
    return np.concatenate(dst), np.concatenate(src)


@pytest.fixture()
def no_batched_flat():
    """the batched FLAT form (round 4) takes precedence where it applies; these walks are about the forms behind it"""
    S.set_option("flatb", 0)
    yield
    S.set_option("flatb", 1)


def test_two_sided_flat_plans_move_every_element_exactly_once(no_batched_flat):
    """Planner check without a GPU: for a spread of shapes / permutations that take the two-sided FLAT form, walk all tiles with the
    plan's own numbers (runs, tiles, offset tables) and compare with the definition -- destination offset sum(i_d * s0_d) receives input
    offset sum(i_d * s1_d) for every index of the box, once."""
    rng = np.random.default_rng(2026)
    shapes = [(5, 60, 50, 7), (17, 9, 33, 31), (3, 100, 90, 3), (7, 30, 40, 9), (6, 16, 16, 16, 5), (12, 10, 14, 9, 11), (10, 50, 60, 10), (31, 65, 33, 17),
              (5, 3000, 7), (3, 40, 50, 3, 9), (24, 30, 30, 20), (9, 11, 700), (13, 6, 900, 2),
              (100, 90, 80), (257, 129, 65), (70, 100, 33), (17, 200, 90)]   # round 4: long unit-stride dims cut evenly (R = 1, p = the lead)
    seen = shared = ragged = 0
    for shape in shapes:
        n = len(shape)
        perms = [tuple(reversed(range(n))), (n - 1,) + tuple(range(1, n - 1)) + (0,)] + [tuple(int(i) for i in rng.permutation(n)) for _ in range(6)]
        for q in perms:
            for dt in (np.float64, np.float32, np.complex128):
                a = S.StridedView(np.zeros(shape, dtype=dt, order="F"))
                b = S.StridedView(np.zeros(tuple(shape[i] for i in q), dtype=dt, order="F"))
                plan = S.make_plan(lambda x: x, None, None, b.size, (b, a.permutedims(q)))
                fr = plan.flat_runs()
                if fr is None:
                    continue
                assert "two-sided" in plan.describe()
                seen += 1
                shared += fr["shared"]
                dims = fr["dims"]
                ragged += any(fr["p"][t] >= 0 and dims[fr["p"][t]] % fr["TP"][t] for t in (0, 1))
                assert fr["R"][0] * fr["TP"][0] <= (512 if fr["shared"] else 128) and fr["R"][1] * fr["TP"][1] <= 128
                # the two runs share no dim
                g0 = {d for d in range(fr["N"]) if fr["in0"][d]} | ({fr["p"][0]} - {-1})
                g1 = {d for d in range(fr["N"]) if fr["in1"][d]} | ({fr["p"][1]} - {-1})
                assert not (g0 & g1), (shape, q, fr)
                idx = np.indices(dims).reshape(len(dims), -1).astype(np.int64)
                want_d = (idx * np.array(fr["s0"], dtype=np.int64)[:, None]).sum(0)
                want_s = (idx * np.array(fr["s1"], dtype=np.int64)[:, None]).sum(0)
                got_d, got_s = _walk_two_sided_flat(fr)
                assert got_d.size == want_d.size, (shape, q, np.dtype(dt).name, fr)
                o1, o2 = np.argsort(got_d, kind="stable"), np.argsort(want_d, kind="stable")
                assert np.array_equal(got_d[o1], want_d[o2]), (shape, q, np.dtype(dt).name, fr)      # every destination element once
                assert np.array_equal(got_s[o1], want_s[o2]), (shape, q, np.dtype(dt).name, fr)      # ... from the element it is the image of
    assert seen >= 60 and shared >= 5 and ragged >= 20, (seen, shared, ragged)


def _walk_one_sided_flat(fs, es):
    """CPU model of smr_k_flat.hip's flat_map_body + its launcher (plain, fused and shared-lead forms): every (destination offset,
    offset in operand kt) pair the launch touches."""
    dims, R, p, q = fs["dims"], fs["R"], fs["p"], fs["q"]
    sflat, sline = (fs["s0"], fs["s1"]) if fs["dir"] == 0 else (fs["s1"], fs["s0"])
    vmax = max(1, 16 // es)
    TQ = 1 << fs["tqlog"]
    nt = (dims[q] + TQ - 1) // TQ
    TQ = (dims[q] + nt - 1) // nt
    TQ = (TQ + vmax - 1) // vmax * vmax
    TP = 1 << fs["tplog"]
    dimp = dims[p] if p >= 0 else 1
    slp = sline[p] if p >= 0 else 0
    sfq = sflat[q]
    ntp, ntq = (dimp + TP - 1) // TP, (dims[q] + TQ - 1) // TQ
    outer = [d for d in range(fs["N"]) if d != p and d != q and not fs["ingroup"][d]]
    roff = np.array(fs["roff"], dtype=np.int64)
    fl, ln = [], []
    for oidx in itertools.product(*[range(dims[d]) for d in outer]):
        bf0 = sum(i * sflat[d] for i, d in zip(oidx, outer))
        bl0 = sum(i * sline[d] for i, d in zip(oidx, outer))
        for tp in range(ntp):
            for tq in range(ntq):
                q0, p0 = tq * TQ, tp * TP
                bf = bf0 + q0 * sfq + p0 * R
                bl = bl0 + q0 * (R if fs["lshare"] else 1) + p0 * slp
                nq = min(TQ, dims[q] - q0)
                nj = min(TP, dimp - p0) * R
                j = np.arange(nj, dtype=np.int64)[:, None]
                x = np.arange(nq, dtype=np.int64)[None, :]
                r, jp = j % R, j // R
                fl.append((bf + x * sfq + j).ravel())
                if fs["lshare"]:
                    ln.append((bl + jp * slp + x * R + r).ravel())
                else:
                    ln.append((bl + roff[r] + jp * slp + x).ravel())
    fl, ln = np.concatenate(fl), np.concatenate(ln)
    return (fl, ln) if fs["dir"] == 0 else (ln, fl)


def test_one_sided_flat_plans_move_every_element_exactly_once(no_batched_flat):
    """The same planner check for the one-sided FLAT forms: plain (either operand flat), fused (planar <-> interleaved) and shared-lead
    (transposition of R-element groups), ragged tiles along both tiled dims, outer dims."""
    rng = np.random.default_rng(7)
    shapes = [(640, 480, 3), (3, 480, 640), (3, 100, 70, 5), (5, 33, 200), (10, 3, 100, 3, 10), (6, 50, 41, 9), (3, 64, 1000), (7, 7, 300), (12, 40, 130),
              (3, 130, 96), (5, 1000, 30), (100, 3, 100, 3), (9, 500, 40)]
    seen = collections.Counter()
    for shape in shapes:
        n = len(shape)
        perms = {tuple(reversed(range(n))), (0,) + tuple(reversed(range(1, n))), tuple(range(1, n)) + (0,), (n - 1,) + tuple(range(n - 1))}
        perms |= {tuple(int(i) for i in rng.permutation(n)) for _ in range(5)}
        for qperm in sorted(perms):
            for dt in (np.float64, np.float32, np.complex128):
                a = S.StridedView(np.zeros(shape, dtype=dt, order="F"))
                b = S.StridedView(np.zeros(tuple(shape[i] for i in qperm), dtype=dt, order="F"))
                plan = S.make_plan(lambda x: x, None, None, b.size, (b, a.permutedims(qperm)))
                fs = plan.flat_side()
                if fs is None:
                    continue
                seen["fuse" if fs["fuse"] else ("lshare" if fs["lshare"] else "plain%d" % fs["dir"])] += 1
                dims = fs["dims"]
                idx = np.indices(dims).reshape(len(dims), -1).astype(np.int64)
                want_d = (idx * np.array(fs["s0"], dtype=np.int64)[:, None]).sum(0)
                want_s = (idx * np.array(fs["s1"], dtype=np.int64)[:, None]).sum(0)
                got_d, got_s = _walk_one_sided_flat(fs, np.dtype(dt).itemsize)
                msg = (shape, qperm, np.dtype(dt).name, plan.describe())
                assert got_d.size == want_d.size, msg
                o1, o2 = np.argsort(got_d, kind="stable"), np.argsort(want_d, kind="stable")
                assert np.array_equal(got_d[o1], want_d[o2]), msg
                assert np.array_equal(got_s[o1], want_s[o2]), msg
    assert seen["plain0"] >= 10 and seen["plain1"] >= 10 and seen["fuse"] >= 5 and seen["lshare"] >= 3, dict(seen)


def test_batched_flat_plans_move_every_element_exactly_once():
    """The batched FLAT form (round 4): blocks of P elements contiguous on both sides.  Walk every workgroup's chunk with the plan's own
    numbers (block size, blocks per workgroup, the srcoff table) exactly as csrc/smr_k_flat.hip:flatb_body does and compare with the
    definition -- destination offset sum(i_d * s0_d) receives input offset sum(i_d * s1_d), once."""
    shapes = [((9, 11, 3000), (1, 0, 2)), ((5, 9, 4001), (1, 0, 2)), ((17, 23, 700), (1, 0, 2)), ((3, 4, 5, 2000), (2, 0, 1, 3)), ((3, 4, 5, 2000), (1, 2, 0, 3)),
              ((7, 6, 333, 9), (1, 0, 2, 3)), ((2, 2, 2, 2, 5000), (3, 1, 2, 0, 4)), ((16, 16, 999), (1, 0, 2)), ((3, 100, 70, 5), (1, 0, 2, 3)),
              ((9, 11, 300, 300), (1, 0, 3, 2)), ((31, 29, 1000), (1, 0, 2)), ((9, 11, 70000), (0, 1, 2)), ((5, 9, 300, 300), (1, 0, 3, 2)),
              ((4, 8, 60, 50, 7), (1, 0, 4, 2, 3)), ((13, 5, 40, 30), (1, 0, 3, 2))]
    seen = 0
    for shape, q in shapes:
        for dt in (np.float64, np.float32, np.complex128):
            a = S.StridedView(np.zeros(shape, dtype=dt, order="F"))
            b = S.StridedView(np.zeros(tuple(shape[i] for i in q), dtype=dt, order="F"))
            plan = S.make_plan(lambda x: x, None, None, b.size, (b, a.permutedims(q)))
            fb = plan.flat_batched()
            es = np.dtype(dt).itemsize
            if fb is None:
                # outside the form: the identity, a block over 4 KiB / 512 elements, or input blocks under 256 bytes that are not adjacent
                permuted_grid = tuple(q[2:]) != tuple(range(2, len(q)))
                pow2 = all(n & (n - 1) == 0 for n in shape[:2])           # power-of-two blocks stay with TILED
                assert (q == tuple(range(len(q)))) or pow2 or shape[0] * shape[1] * es > 4096 or shape[0] * shape[1] > 512 or \
                    (permuted_grid and shape[0] * shape[1] * es < 256), (shape, q, dt)
                continue
            seen += 1
            g, P, K, N = fb["g"], fb["P"], fb["K"], fb["N"]
            dims, s0, s1 = fb["dims"], np.array(fb["s0"], dtype=np.int64), np.array(fb["s1"], dtype=np.int64)
            assert P == int(np.prod(dims[:g])) and P * es <= 4096 and P <= 512 and s0[g] == P and s1[g] >= P
            idx = np.indices(dims).reshape(N, -1).astype(np.int64)
            want_d, want_s = (idx * s0[:, None]).sum(0), (idx * s1[:, None]).sum(0)
            nb = dims[g]
            outer = np.indices(dims[g + 1:]).reshape(N - g - 1, -1).astype(np.int64) if N > g + 1 else np.zeros((0, 1), dtype=np.int64)
            obase = (outer * s0[g + 1:, None]).sum(0) if N > g + 1 else np.zeros(1, dtype=np.int64)
            obase1 = (outer * s1[g + 1:, None]).sum(0) if N > g + 1 else np.zeros(1, dtype=np.int64)
            srcoff = np.array(fb["srcoff"], dtype=np.int64)
            got_d, got_s = [], []
            for ch in range((nb + K - 1) // K):
                n = min(K, nb - ch * K) * P
                t = np.arange(n, dtype=np.int64)
                blk, r = t // P, t % P
                d_local = ch * K * P + t                    # destination: the chunk's elements in memory order
                s_local = (ch * K + blk) * s1[g] + srcoff[r]  # input: the same block (wherever it sits), position srcoff[r]
                got_d.append((obase[:, None] + d_local[None, :]).ravel())
                got_s.append((obase1[:, None] + s_local[None, :]).ravel())
            got_d, got_s = np.concatenate(got_d), np.concatenate(got_s)
            msg = (shape, q, np.dtype(dt).name, plan.describe())
            assert got_d.size == want_d.size, msg
            o1, o2 = np.argsort(got_d, kind="stable"), np.argsort(want_d, kind="stable")
            assert np.array_equal(got_d[o1], want_d[o2]), msg
            assert np.array_equal(got_s[o1], want_s[o2]), msg
    assert seen >= 18, seen


def test_round4_planner_rules_for_ragged_and_batched_shapes():
    """Which family / form the planner picks for the shapes round 4 measured (profiles/r04_flat2_long_ab.txt, r04_flat2_batched.txt,
    r04_perf_sanity.txt) -- so that a change of these rules is a deliberate one."""
    def desc(shape, q, dt=np.float64, f=lambda x: x):
        a = S.StridedView(np.zeros(shape, dtype=dt, order="F"))
        b = S.StridedView(np.zeros(tuple(shape[i] for i in q), dtype=dt, order="F"))
        return S.make_plan(f, None, None, b.size, (b, a.permutedims(q))).describe()

    # long unit-stride dims: evenly cut leads when 32 x 32 tiles would be poorly filled (and the array has >= 8 MiB) ...
    assert "two-sided" in desc((257, 129, 65), (2, 1, 0)) and "dest_run=1x" in desc((257, 129, 65), (2, 1, 0))
    # ... or when an extent is not a multiple of the 16-byte vector length (TILED would move single elements as well) ...
    # (round 6: only from 32 MiB on -- below, TILED keeps 16-byte accesses at element alignment and is ahead)
    assert "two-sided" in desc((2049, 2051), (1, 0)) and "two-sided" in desc((2899, 2901), (1, 0), np.float32)
    assert "family=tiled" in desc((999, 1001), (1, 0)) and "family=tiled" in desc((999, 1001), (1, 0), np.float32)
    # ... or next to a short lead that is not a power of two
    assert "two-sided" in desc((17, 33, 65, 31), (3, 2, 0, 1))
    # well-filled, even and power-of-two shapes stay with TILED; so do small arrays
    for shape, q in (((1400, 1500), (1, 0)), ((4000, 4100), (1, 0)), ((4096, 4096), (1, 0)), ((128, 128, 64), (1, 0, 2)), ((100, 90, 80), (1, 0, 2)),
                     ((64, 2, 64, 2, 16), (3, 0, 1, 2, 4)), ((2, 2, 256, 2, 2, 256), (5, 4, 3, 2, 1, 0))):
        assert "family=tiled" in desc(shape, q), (shape, q, desc(shape, q))
    # contiguous small blocks with a non-power-of-two extent: the batched form, also with a permuted batch grid (blocks of >= 256 bytes)
    assert "batched block=99" in desc((9, 11, 3000), (1, 0, 2)) and "batched block=99" in desc((9, 11, 70, 60), (1, 0, 3, 2))
    assert "batched" in desc((100, 3, 100, 3, 10), (1, 0, 4, 3, 2))
    assert "batched" not in desc((2, 128, 2, 128, 8), (2, 0, 1, 3, 4)) and "batched" not in desc((16, 16, 999), (1, 0, 2))   # power-of-two blocks: TILED
    assert "batched" not in desc((5, 9, 300, 300), (1, 0, 3, 2), np.float32)                                                  # 180-byte blocks on a permuted grid
    # a stepped range behind a permutation: its smallest stride (3) is the input's near-unit axis
    A = S.StridedView(np.zeros((2048, 2048), order="F"))
    B = S.StridedView(np.zeros((2048, 2048), order="F"))
    d = S.make_plan(lambda x, y: x + y, None, None, (682, 682),
                    (B.sview(slice(0, 682), slice(0, 682)), A.sview(slice(0, 2046, 3), slice(0, 682)), A.permutedims((1, 0)).sview(slice(0, 682), slice(0, 2046, 3)))).describe()
    assert "family=tiled" in d, d


def test_round6_planner_rules():
    """Round 6: the ORBIT work list has a PAIR form for 4^4 cubes of 8-byte elements (two orbits per workgroup: half as many
    workgroups, every one of them a whole number of orbits), not for Float32 or 8^4 cubes; COL reductions size their lane map to
    the row when that fills the workgroup better; odd extents below 32 MiB stay in TILED (element-aligned vectors)."""
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]

    def sum4(n, dt):
        a = S.StridedView(np.zeros((n,) * 4, dtype=dt, order="F"))
        c = S.StridedView(np.zeros((n,) * 4, dtype=dt, order="F"))
        return S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, a.size, (c,) + tuple(a.permutedims(p) for p in perms)).describe()

    d = sum4(32, np.float64)
    assert "family=orbit" in d and "tile=d0:4,d1:4,d2:4,d3:4" in d and "grid=1024" in d and "pair_grid=512" in d, d
    assert "pair_grid=512" in sum4(32, np.complex64)
    assert "pair_grid=" in sum4(24, np.float64) and "pair_grid=" in sum4(36, np.float64)   # 6 and 9 tiles per dim (an odd one out is fine)
    assert "pair_grid=" not in sum4(32, np.float32)                                        # 4-byte elements: one orbit per workgroup
    assert "tile=d0:8" in sum4(64, np.float64) and "pair_grid=" not in sum4(64, np.float64)  # 8^4 cubes
    S.set_option("orbit_pair", 0)
    try:
        assert "pair_grid=" not in sum4(32, np.float64)
    finally:
        S.set_option("orbit_pair", 1)

    def red(dims, rd, dt=np.float32):
        a = S.StridedView(np.zeros(dims, dtype=dt, order="F"))
        out = a.similar(size=tuple(1 if i in rd else n for i, n in enumerate(dims)))
        return S.make_plan(lambda x: x, "+", "zero", dims, S.promoteshape(dims, out, a)).describe()

    assert "lanes=25x10" in red((100, 90, 80, 7), (1, 2, 3)) and "lanes=50x5" in red((100, 90, 80, 7), (1, 2, 3), np.float64)
    assert "lanes=" not in red((128, 90, 80), (1, 2))             # 128 Float32 = 32 vectors: the power of two is exact
    assert "lanes=24x10" in red((96, 90, 80, 7), (1, 2, 3))       # 24 vectors: one segment of 24 lanes (94 %), not two of 12
    assert "lanes=" not in red((100, 90, 80, 7), (3,))            # a reduction of 7 rows keeps "one lane walks them"
    # a narrower power-of-two segment that puts a workgroup on every CU beats a split + second pass
    assert "form=col lanes_per_out=32 split=1" in red((100, 90, 80, 7), (2, 3))
    S.set_option("reduce_col_exact", 0)
    try:
        assert "lanes=" not in red((100, 90, 80, 7), (1, 2, 3))
    finally:
        S.set_option("reduce_col_exact", 1)


def test_orbit_pair_work_list_covers_every_tile_once():
    """The PAIR form's work list (smr_plan_orbit_pairs): eight tiles per workgroup; every tile of the box exactly once; each slot set is
    an orbit under the cyclic shift of the tile coordinates (slot g = the shift applied g times to slot 0); and in nearly all
    workgroups set 1's slot 0 is the unit-axis neighbour of set 0's (that is what makes 64-byte runs)."""
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    for n in (32, 24, 36):
        a = S.StridedView(np.zeros((n,) * 4, dtype=np.float64, order="F"))
        c = S.StridedView(np.zeros((n,) * 4, dtype=np.float64, order="F"))
        plan = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, a.size, (c,) + tuple(a.permutedims(p) for p in perms))
        lst = plan.orbit_pairs()
        nt = n // 4
        assert lst and len(lst) % 64 == 0, (n, len(lst))           # whole workgroups, a multiple of 8 of them (one run per XCD)
        seen = np.zeros(nt ** 4, dtype=np.int64)
        paired = live = 0

        def coords(t):
            return [(t // nt ** d) % nt for d in range(4)]

        def tid(cs):
            return sum(cc * nt ** d for d, cc in enumerate(cs))

        for w in range(len(lst) // 8):
            sets = [lst[w * 8 + b * 4:w * 8 + b * 4 + 4] for b in range(2)]
            if sets[0][0] == 0xffffffff:
                continue
            live += 1
            for b, st in enumerate(sets):
                if b == 1 and st == sets[0]:
                    continue                                          # an odd set out runs twice
                for t in set(st):
                    seen[t] += 1
                # closed under the shift: with every tile a set holds all its cyclic shifts (one orbit of four tiles, or -- tiles on
                # a diagonal -- several shorter orbits sharing the set)
                for t in st:
                    c0 = coords(t)
                    assert {tid(c0[k:] + c0[:k]) for k in range(4)} <= set(st), (n, w, b, st)
            if sets[1][0] == sets[0][0] + 1 and coords(sets[0][0])[0] % 2 == 0:
                paired += 1
        assert np.all(seen == 1), (n, int((seen != 1).sum()))
        assert paired >= 0.85 * live, (n, paired, live)
