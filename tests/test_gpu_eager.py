"""GPU: eager direct dispatch on a library-owned stream (smr_stream_create; csrc/smr_seq.cpp).  The library submits every launch itself
as an AQL packet on one of four HSA queues, chosen by the data dependencies: independent executions run concurrently, conflicting
ones are ordered (same queue, or a barrier-AND packet across queues).  The contract under test is the reference's
(/root/reference/src/mapreduce.jl:203-223: spawn what is independent, wait where it must): whatever overlaps, the results are
those of executing the calls one after the other.  Truth: NumPy, applied sequentially."""
import ctypes as C

import numpy as np
import pytest

import strided_jl_amd as S
from strided_jl_amd import _lib as L

pytestmark = pytest.mark.gpu


def dev(arr):
    import torch
    a = np.asfortranarray(arr)
    t = torch.from_numpy(a.ravel(order="F").copy()).cuda()
    st, s = [], 1
    for d in a.shape:
        st.append(s)
        s *= d
    return S.StridedView(t, a.shape, tuple(st), 0)


def host(view):
    return view.parent.cpu().numpy().reshape(view.size, order="F")


@pytest.fixture()
def stream():
    import torch
    torch.cuda.synchronize()
    st = S.Stream()
    yield st
    st.close()


def stats():
    return {k: S.get_option("eager_" + k) for k in ("launches", "free", "same", "cross", "fallback")}


def test_direct_dispatch_is_used_and_correct(stream):
    import torch
    rng = np.random.default_rng(1)
    a = rng.standard_normal((48, 40, 36))
    A, B, C = dev(a), dev(np.zeros((36, 40, 48))), dev(np.zeros((36, 40, 48)))
    torch.cuda.synchronize()
    before = stats()
    p1 = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims((2, 1, 0))))
    p2 = S.make_plan(lambda x: x * 2 + 1, None, None, C.size, (C, B))          # read-after-write on B
    p1.execute(stream.handle)
    p2.execute(stream.handle)
    stream.synchronize()
    after = stats()
    assert after["launches"] - before["launches"] == 2, (before, after)
    at = np.transpose(a, (2, 1, 0))
    assert np.array_equal(host(B), at) and np.array_equal(host(C), at * 2 + 1)


@pytest.mark.parametrize("seed", range(6))
def test_random_programs_equal_sequential_numpy(stream, seed):
    """Random programs over a pool of buffers: maps, transposes, in-place updates, accumulating partial reductions -- with every
    kind of hazard (read-after-write, write-after-read, write-after-write, several queues feeding one consumer)."""
    import torch
    rng = np.random.default_rng(100 + seed)
    n = 72   # 3 MB per buffer: a launch lasts longer than its submission, so that hazards are met with work still in flight
    pool_np = [rng.integers(-3, 4, size=(n, n, n)).astype(np.float64) for _ in range(6)]      # small integers: exact in any order
    red_np = [np.zeros((n, 1, n)), np.zeros((1, n, 1))]
    pool = [dev(x) for x in pool_np]
    red = [dev(x) for x in red_np]
    torch.cuda.synchronize()
    before = stats()
    perms = [(0, 1, 2), (1, 0, 2), (2, 1, 0), (1, 2, 0), (0, 2, 1), (2, 0, 1)]
    keep = []
    for step in range(120):
        kind = rng.integers(0, 5)
        d = int(rng.integers(0, 6))
        s1, s2 = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        p = perms[int(rng.integers(0, 6))]
        if kind == 0 and s1 != d:        # dest = permutedims(src)
            plan = S.make_plan(lambda x: x, None, None, pool[d].size, (pool[d], pool[s1].permutedims(p)))
            pool_np[d] = np.transpose(pool_np[s1], p).copy()
        elif kind == 1 and s1 != d and s2 != d:   # dest = src1 - permutedims(src2)
            plan = S.make_plan(lambda x, y: x - y, None, None, pool[d].size, (pool[d], pool[s1], pool[s2].permutedims(p)))
            pool_np[d] = pool_np[s1] - np.transpose(pool_np[s2], p)
        elif kind == 2:                  # in place: dest = min(dest, 5) * -1 (stays small)
            plan = S.make_plan(lambda x: -S.fn.min(x, 5.0), None, None, pool[d].size, (pool[d], pool[d]))
            pool_np[d] = -np.minimum(pool_np[d], 5.0)
        elif kind == 3:                  # accumulate a partial reduction INTO a small destination (reads it, too)
            r = int(rng.integers(0, 2))
            from strided_jl_amd.broadcast import promoteshape
            plan = S.make_plan(lambda x: x, "+", None, pool[s1].size, promoteshape(pool[s1].size, red[r], pool[s1]))
            axes = (1,) if r == 0 else (0, 2)
            red_np[r] = red_np[r] + pool_np[s1].sum(axis=axes, keepdims=True)
        elif kind == 4 and s1 != d:      # dest = clamp(src + broadcast of a reduction result)
            r = int(rng.integers(0, 2))
            from strided_jl_amd.broadcast import promoteshape
            plan = S.make_plan(lambda x, y: S.fn.max(S.fn.min(x + y, 4.0), -4.0), None, None, pool[d].size,
                               promoteshape(pool[d].size, pool[d], pool[s1], red[r]))
            pool_np[d] = np.maximum(np.minimum(pool_np[s1] + red_np[r], 4.0), -4.0)
        else:
            continue
        keep.append(plan)
    for plan in keep:          # submitted back to back: ~2 us of host time each
        plan.execute(stream.handle)
    stream.synchronize()
    after = stats()
    for i in range(6):
        assert np.array_equal(host(pool[i]), pool_np[i]), ("buffer", i)
    for r in range(2):
        assert np.array_equal(host(red[r]), red_np[r]), ("reduction", r)
    used = {k: after[k] - before[k] for k in after}
    assert used["launches"] >= 60 and used["fallback"] == 0, used
    assert used["same"] + used["cross"] >= 10, used           # hazards were met with work in flight


def test_copies_and_scalar_results_are_ordered_with_direct_launches(stream):
    import torch
    lib = L.load()
    rng = np.random.default_rng(3)
    n = 1 << 16
    x = rng.standard_normal(n)
    t = torch.zeros(n, dtype=torch.float64, device="cuda")
    out = torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    X, O = S.StridedView(t, (n,), (1,), 0), S.StridedView(out, (n,), (1,), 0)
    h = C.c_void_p(stream.handle)
    back = np.zeros(n)
    plan = S.make_plan(lambda v: v * 3, None, None, (n,), (O, X))
    for rep in range(3):
        xs = np.ascontiguousarray(x + rep)
        L.check(lib.smr_memcpy_h2d(C.c_void_p(t.data_ptr()), xs.ctypes.data_as(C.c_void_p), n * 8, h))      # HIP copy ...
        plan.execute(stream.handle)                                                                          # ... then a direct launch ...
        L.check(lib.smr_memcpy_d2h(back.ctypes.data_as(C.c_void_p), C.c_void_p(out.data_ptr()), n * 8, h))   # ... then a HIP copy
        stream.synchronize()
        assert np.array_equal(back, xs * 3), rep
    # a complete reduction's scalar result comes back through the same stream
    p, keep = S.build_problem(S.fn.abs2, "+", None, (n,), (S.StridedView(torch.zeros(1, dtype=torch.float64, device="cuda"), (n,), (0,), 0), X),
                              stream=stream.handle)
    res = C.c_double(0)
    L.check(lib.smr_mapreduce_scalar(C.byref(p), C.byref(res)))
    assert abs(res.value - float(((x + 2) ** 2).sum())) <= 1e-9 * float(((x + 2) ** 2).sum())


def test_through_hip_when_switched_off(stream):
    import torch
    rng = np.random.default_rng(4)
    a = rng.standard_normal((64, 64))
    A, B = dev(a), dev(np.zeros((64, 64)))
    torch.cuda.synchronize()
    S.set_option("eager_direct", 0)
    try:
        before = stats()
        S.make_plan(lambda x: x + 1, None, None, B.size, (B, A.permutedims((1, 0)))).execute(stream.handle)
        stream.synchronize()
        assert stats()["launches"] == before["launches"]
    finally:
        S.set_option("eager_direct", 1)
    assert np.array_equal(host(B), a.T + 1)


def test_front_ends_inside_a_stream_block_use_direct_dispatch():
    """`with S.Stream():` -- broadcast, map!, permutedims!, reductions through the Python mirror, submitted by the library itself."""
    import torch
    rng = np.random.default_rng(21)
    a = rng.standard_normal((30, 20, 25))
    b = rng.standard_normal((30, 20, 25))
    A, B = dev(a), dev(b)
    out = dev(np.zeros((25, 20, 30)))
    torch.cuda.synchronize()
    before = stats()
    st = S.Stream()
    with st:
        S.permutedims_(out, A, (2, 1, 0))
        C1 = S.map(lambda x, y: x * y - 2 * x, A, B)                       # allocates (torch), launches on the stream
        S.map_(lambda x: x + 1, out, out)                                  # depends on the permutedims! above
        r = S.sum(C1, dims=(0, 2))
        big = dev(rng.standard_normal((600, 500, 8)))
        torch.cuda.synchronize()                                           # (torch's upload is not ordered against the library stream)
        inside = S.sum(big, dims=(0, 1)).toarray()                         # toarray() inside the block finishes the library stream first
    torch.cuda.synchronize()
    after = stats()
    st.close()
    assert np.allclose(inside, host(big).sum(axis=(0, 1), keepdims=True), rtol=1e-12)
    assert after["launches"] - before["launches"] >= 4, (before, after)
    assert np.array_equal(host(out), np.transpose(a, (2, 1, 0)) + 1)
    assert np.array_equal(C1.toarray(), a * b - 2 * a)
    assert np.allclose(r.toarray(), (a * b - 2 * a).sum(axis=(0, 2), keepdims=True), rtol=1e-12)


def test_scratch_plan_alternating_between_a_library_stream_and_hip_streams(stream):
    """A split partial reduction owns partials + arrival counters: its executions must never overlap.  On a library-owned stream the
    launches are not HIP work (no event covers them), so alternating with ordinary HIP streams is ordered on the host (csrc/smr_api.cpp:
    execute_owned)."""
    import torch
    from strided_jl_amd.broadcast import promoteshape
    rng = np.random.default_rng(77)
    a = rng.integers(-4, 5, size=(33, 200000)).astype(np.float64)
    A = dev(a)
    out = dev(np.zeros((33, 1)))
    torch.cuda.synchronize()
    plan = S.make_plan(lambda x: x, "+", None, A.size, promoteshape(A.size, out, A))     # accumulates INTO out
    side = torch.cuda.Stream()
    want = a.sum(axis=1, keepdims=True)
    n = 0
    for rep in range(6):
        plan.execute(stream.handle)                 # direct dispatch
        plan.execute(side.cuda_stream)              # HIP, another stream
        plan.execute(stream.handle)
        plan.execute(0)                             # HIP, the null stream
        n += 4
    stream.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(host(out), want * n)
