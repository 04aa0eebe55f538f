"""GPU: round-5 changes to the library's own dispatch path (csrc/smr_seq.cpp, csrc/smr_kmeta.cpp).

  * hidden kernel arguments at the offsets the code object's metadata names, confirmed by a self-test packet;
  * smr_seq_run returns after the doorbells; smr_seq_wait / the stream waits (4b);
  * acquire fences only on packets that read what the sequence writes; the heaviest chain cut in two (item 2);
  * pending HIP work tracked per library-owned stream;
  * a failed direct path is reported and everything goes through HIP.
Truth everywhere: NumPy, applied in recorded order (the reference's contract: src/mapreduce.jl:203-223, results of sequential execution).
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import strided_jl_amd as S
from strided_jl_amd import _lib as L

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PERMS = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]


def dview(arr):
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


def cur():
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def field(info, key):
    for tok in info.split():
        if tok.startswith(key + "="):
            return tok.split("=", 1)[1]
    raise KeyError(key + " not in: " + info)


def step_plans(n, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n,) * 4)
    A, B, Cc = dview(a), dview(np.zeros_like(a)), dview(np.zeros_like(a))
    p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    p3 = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in PERMS))
    want2 = np.transpose(a, (3, 2, 1, 0))
    want3 = ((np.transpose(a, PERMS[0]) + np.transpose(a, PERMS[1])) + np.transpose(a, PERMS[2])) + np.transpose(a, PERMS[3])
    return a, (A, B, Cc), (p2, p3), (want2, want3)


def test_kernarg_layout_comes_from_metadata_and_the_selftest_passed():
    import torch
    _, (A, B, Cc), (p2, p3), (w2, w3) = step_plans(16)
    q = S.Sequence().add(p2).add(p3)
    q.run(2, cur())
    q.wait()
    torch.cuda.synchronize()
    info = q.info()
    assert field(info, "backend") == "aql", info
    assert field(info, "kernarg_layout") == "metadata+verified", info
    assert field(info, "agent") == "pci-address", info     # the HSA agent is the HIP device's, found by PCI domain / bus / device (matters with 8 GPUs)
    assert np.array_equal(host(B), w2) and np.array_equal(host(Cc), w3)


def test_acquire_only_where_the_sequence_reads_what_it_writes():
    """bench step: nobody writes A -> no inner packet acquires; a chain B = A', C = B + B' reads B -> its packet acquires"""
    import torch
    _, (A, B, Cc), (p2, p3), (w2, w3) = step_plans(32, 1)
    q = S.Sequence().add(p2).add(p3)
    q.run(7, cur())
    q.wait()
    torch.cuda.synchronize()
    info = q.info()
    assert field(info, "acquire").startswith("by-need(0"), info
    assert np.array_equal(host(B), w2) and np.array_equal(host(Cc), w3)
    rng = np.random.default_rng(2)
    a = rng.standard_normal((160, 160))
    A2, B2, C2 = dview(a), dview(np.zeros_like(a)), dview(np.zeros_like(a))
    acc = dview(np.zeros_like(a))
    r1 = S.make_plan(lambda x: x, None, None, A2.size, (B2, A2.permutedims((1, 0))))
    r2 = S.make_plan(lambda x, y: x + y, None, None, A2.size, (C2, B2, B2.permutedims((1, 0))))   # reads what r1 wrote
    r3 = S.make_plan(lambda x, y: x + y, None, None, A2.size, (acc, acc, C2))                     # in place: acc += C (replays accumulate)
    q2 = S.Sequence().add(r1).add(r2).add(r3)
    q2.run(5, cur())
    q2.wait()
    torch.cuda.synchronize()
    info2 = q2.info()
    assert field(info2, "acquire").startswith("by-need(2"), info2   # r2 and r3; r1 reads only A
    assert np.array_equal(host(C2), a.T + a) and np.array_equal(host(acc), 5 * (a.T + a))


@pytest.mark.parametrize("n", [16, 32])
def test_chains_are_cut_by_default_and_results_are_those_of_in_order_execution(n):
    import torch
    _, (A, B, Cc), (p2, p3), (w2, w3) = step_plans(n, 3)
    cut = "3" if n == 32 else "2"   # (a launch of fewer than 128 workgroups is not cut: 16^4 has 72 orbits / 64 tiles)
    for lay, want_q in ((None, None), ({"slices": 1}, "2"), ({"slices:1": 2, "queues": 3}, cut), ({"slices:0": 2, "queues": 3}, cut)):
        q = S.Sequence().add(p2).add(p3)
        for k, v in (lay or {}).items():
            q.set(k, v)
        B.parent.zero_(); Cc.parent.zero_()
        torch.cuda.synchronize()
        q.run(4, cur())
        q.wait()
        torch.cuda.synchronize()
        info = q.info()
        if want_q:
            assert field(info, "queues") == want_q, info
        else:   # automatic: every launch is self-released (no release fence) -> every chain that can be cut is cut, up to 4 queues
            assert field(info, "self_released") == field(info, "packets"), info
            if n == 32:
                assert field(info, "queues") == "4" and field(info, "sliced") == "2", info
            else:
                assert field(info, "queues") == "2" and field(info, "sliced") == "0", info
        assert np.array_equal(host(B), w2) and np.array_equal(host(Cc), w3), info


def test_first_acquire_at_agent_scope_sees_dma_uploads_and_falls_back_to_system_for_host_memory():
    """A replay whose inputs are device-local starts with an agent-scope acquire; the inputs stay in the L2s across replays (no inner
    acquire), so this is the case where a stale line would show: upload new contents by DMA between replays and compare."""
    import torch
    n = 32
    a = np.random.default_rng(7).standard_normal((n,) * 4)
    A, B, Cc = dview(a), dview(np.zeros_like(a)), dview(np.zeros_like(a))
    p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    p3 = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (Cc,) + tuple(A.permutedims(p) for p in PERMS))
    q = S.Sequence().add(p2).add(p3)
    st = S.Stream()
    try:
        q.run(3, st.handle); q.wait()
        assert field(q.info(), "first_acquire") == "agent", q.info()
        pin = torch.empty(n ** 4, dtype=torch.float64).pin_memory()
        for rep in range(5):
            new = a * (rep + 2) + rep
            pin.copy_(torch.from_numpy(new.ravel(order="F").copy()))
            A.parent.copy_(pin, non_blocking=True)       # DMA into device memory the L2s hold from the previous replay
            torch.cuda.synchronize()
            q.run(3, st.handle); q.wait()
            torch.cuda.synchronize()
            assert np.array_equal(host(B), np.transpose(new, (3, 2, 1, 0))), rep
            want = ((np.transpose(new, PERMS[0]) + np.transpose(new, PERMS[1])) + np.transpose(new, PERMS[2])) + np.transpose(new, PERMS[3])
            assert np.array_equal(host(Cc), want), rep
        # an input in pinned host memory (zero-copy, addressed through the C ABI by its raw pointer): system scope
        hp = torch.from_numpy(np.arange(4096.0)).pin_memory()
        out = torch.zeros(4096, dtype=torch.float64, device="cuda")
        O = S.StridedView(out, (4096,), (1,), 0)
        X = S.StridedView(torch.zeros(4096, dtype=torch.float64, device="cuda"), (4096,), (1,), 0)
        ph = S.make_plan(lambda x: x * 2, None, None, (4096,), (O, X))
        qh = S.Sequence().add(ph, bases=[out.data_ptr(), hp.data_ptr()])   # rebind the input to the pinned host buffer
        qh.run(1, st.handle); qh.wait()
        info = qh.info()
        if field(info, "backend") == "aql":
            assert field(info, "first_acquire") == "system", info
        for rep in range(3):
            hp += 1.0                                    # the host writes the input between replays
            qh.run(1, st.handle); qh.wait()
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), (np.arange(4096.0) + rep + 1) * 2), rep
    finally:
        st.close()


def test_run_returns_before_the_replay_completes_on_a_library_stream():
    """asynchronous smr_seq_run: the host gets the thread back while the device works; wait() is separate"""
    import torch
    _, (A, B, Cc), (p2, p3), (w2, w3) = step_plans(32, 4)
    st = S.Stream()
    try:
        q = S.Sequence().add(p2).add(p3)
        q.run(2, st.handle); q.wait()
        K = 4000                       # ~22 ms of device time
        t0 = time.perf_counter()
        q.run(K, st.handle)
        t_run = time.perf_counter() - t0
        acc = 0.0
        for i in range(200):           # the host does other work meanwhile
            acc += float(np.dot(np.arange(64.0), np.arange(64.0)))
        q.wait()
        t_all = time.perf_counter() - t0
        assert t_run < 0.5 * t_all, (t_run, t_all)   # returning took a fraction of the replay
        assert t_all > 0.010, t_all                  # the replay really ran for ~K x 5.5 us
        st.synchronize()
        assert np.array_equal(host(B), w2) and np.array_equal(host(Cc), w3)
        # a launch on the same library stream issued right after an asynchronous run is ordered behind it
        Z = dview(np.zeros((32,) * 4))
        pz = S.make_plan(lambda x, y: x - y, None, None, A.size, (Z, Cc, B))
        B.parent.zero_(); Cc.parent.zero_()
        torch.cuda.synchronize()
        q.run(50, st.handle)
        pz.execute(st.handle)
        st.synchronize()
        assert np.array_equal(host(Z), w3 - w2)
    finally:
        st.close()


def test_work_queued_on_a_hip_stream_after_an_asynchronous_run_sees_the_results():
    """on a HIP stream a holding kernel (or hipStreamWaitValue64) keeps later stream work behind the replay"""
    import torch
    _, (A, B, Cc), (p2, p3), (w2, w3) = step_plans(32, 5)
    hs = torch.cuda.Stream()
    q = S.Sequence().add(p2).add(p3)
    with torch.cuda.stream(hs):
        q.run(2, int(hs.cuda_stream)); q.wait()
        for rep in range(3):
            B.parent.zero_(); Cc.parent.zero_()
            q.run(300, int(hs.cuda_stream))          # ~1.7 ms; returns after the doorbells
            snap_b = B.parent.clone()                # queued on the stream: must run after the replay
            snap_c = Cc.parent.clone()
            hs.synchronize()
            q.wait()
            assert np.array_equal(snap_b.cpu().numpy().reshape(B.size, order="F"), w2), rep
            assert np.array_equal(snap_c.cpu().numpy().reshape(Cc.size, order="F"), w3), rep
    assert "holding-kernel" in q.info() or "hipStreamWaitValue64" in q.info()
    # blocking mode on request
    q.set("async", 0)
    B.parent.zero_()
    torch.cuda.synchronize()
    q.run(3, cur())
    assert np.array_equal(host(B), w2)               # no wait(), no synchronize: the call itself waited
    q.wait()


def test_two_library_streams_each_drains_its_own_hip_work():
    """a copy queued through HIP on stream A must not be forgotten because a direct launch on stream B drained B"""
    import torch
    lib = L.load()
    n = 1 << 22                                       # 32 MiB per copy: the copy is still in flight when the launches are issued
    sa, sb = S.Stream(), S.Stream()
    try:
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        y = torch.zeros(n, dtype=torch.float64, device="cuda")
        u = torch.ones(4096, dtype=torch.float64, device="cuda")
        v = torch.zeros(4096, dtype=torch.float64, device="cuda")
        pin = torch.empty(n, dtype=torch.float64).pin_memory()
        torch.cuda.synchronize()
        X, Y = S.StridedView(x, (n,), (1,), 0), S.StridedView(y, (n,), (1,), 0)
        U, V = S.StridedView(u, (4096,), (1,), 0), S.StridedView(v, (4096,), (1,), 0)
        pa = S.make_plan(lambda t: t + 1, None, None, (n,), (Y, X))       # on A: reads what the copy writes
        pb = S.make_plan(lambda t: t * 2, None, None, (4096,), (V, U))    # on B: unrelated
        pa.execute(sa.handle); pb.execute(sb.handle)
        sa.synchronize(); sb.synchronize()
        for rep in range(6):
            pin.fill_(float(rep + 1))
            L.check(lib.smr_memcpy_h2d(C.c_void_p(x.data_ptr()), C.c_void_p(pin.data_ptr()), n * 8, C.c_void_p(sa.handle)))   # HIP work on A
            pb.execute(sb.handle)      # a direct launch on B: drains B only
            pa.execute(sa.handle)      # must still wait for A's copy
            sa.synchronize(); sb.synchronize()
            got = y.cpu().numpy()
            assert got[0] == rep + 2 and got[-1] == rep + 2 and np.all(got == rep + 2), rep
    finally:
        sa.close(); sb.close()


def test_self_released_launches_need_no_release_fence_where_plain_stores_do():
    """Two plans write ONE destination through different tilings (a line of B is written from one XCD, then from another).  With
    write-through stores and no release fence on the packets the result is the in-order one, every time; the same packets with
    plain stores and the fence dropped lose lines to a late write-back (reported, not asserted: it is a race)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import waw_hazard
    wrong_self, info = waw_hazard.run(32, "self-released", reps=30)
    assert wrong_self == 0, (wrong_self, info)
    assert field(info, "self_released") == "2" and field(info, "components") == "1", info
    wrong_agent, _ = waw_hazard.run(32, "agent release", reps=10)
    assert wrong_agent == 0
    wrong_plain, _ = waw_hazard.run(32, "plain, no release", reps=30)
    print("plain stores without a release fence: %d wrong elements in 30 runs (the hazard the fence / the write-through stores remove)" % wrong_plain)
    torch.cuda.synchronize()


CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import strided_jl_amd as S
a = np.random.default_rng(0).standard_normal((32, 32, 32))
t = torch.from_numpy(a.ravel(order="F").copy()).cuda()
o = torch.zeros_like(t)
A = S.StridedView(t, a.shape, (1, 32, 1024), 0)
B = S.StridedView(o, a.shape, (1, 32, 1024), 0)
p = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((2, 1, 0))))
q = S.Sequence().add(p)
q.run(3, int(torch.cuda.current_stream().cuda_stream)); q.wait()
torch.cuda.synchronize()
st = S.Stream()
p.execute(st.handle); st.synchronize()
ok = np.array_equal(o.cpu().numpy().reshape(a.shape, order="F"), np.transpose(a, (2, 1, 0)))
print("INFO", q.info())
print("EAGER", S.get_option("eager_launches"))
print("OK" if ok else "WRONG")
"""


def test_a_failed_selftest_switches_the_direct_path_off_and_everything_runs_through_hip():
    env = dict(os.environ, SMR_DIRECT_SELFTEST="fail")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout.split(), r.stdout
    info = [l for l in r.stdout.splitlines() if l.startswith("INFO")][0]
    assert "backend=hip" in info and "self-test" in info, info
    assert [l for l in r.stdout.splitlines() if l.startswith("EAGER")][0].split()[1] == "0", r.stdout
    # and with the v5 rule instead of the metadata ($SMR_DIRECT_METADATA=0) the self-test still confirms the layout
    env = dict(os.environ, SMR_DIRECT_METADATA="0")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout.split(), r.stdout + r.stderr
    info = [l for l in r.stdout.splitlines() if l.startswith("INFO")][0]
    assert "backend=aql" in info and "kernarg_layout=v5-rule+verified" in info, info


@pytest.mark.parametrize("shape,perm,dt", [((128, 64, 64, 128), (3, 2, 1, 0), "float64"), ((256, 64, 32, 128), (3, 2, 1, 0), "float64"),
                                           ((512, 256, 512), (2, 1, 0), "float64"), ((128, 96, 64, 128), (3, 1, 2, 0), "complex128"),
                                           ((128, 128, 32, 128), (3, 0, 1, 2), "int64")])
def test_hbm_sized_transposing_copies_lean_kernel_and_grid_order_are_bit_exact(shape, perm, dt):
    """>= 512 MiB transposing copies of rank >= 3 take the lean 128 x 32 kernel (k_xpose_big) and the planner's grid order; the general
    kernel in canonical order is the comparison; truth = torch."""
    import torch
    n = int(np.prod(shape))
    tdt = getattr(torch, dt)
    if dt == "int64":
        tA = torch.randint(-2 ** 62, 2 ** 62, (n,), dtype=tdt, device="cuda")
    elif dt == "complex128":
        tA = torch.randn(n, dtype=tdt, device="cuda")
    else:
        tA = torch.randn(n, dtype=tdt, device="cuda")
    rank = len(shape)
    oshape = tuple(shape[p] for p in perm)

    def cm(t, sh):
        st, s = [], 1
        for d in sh:
            st.append(s)
            s *= d
        return S.StridedView(t, sh, tuple(st), 0)

    want = tA.reshape(tuple(reversed(shape))).permute(*[rank - 1 - perm[rank - 1 - i] for i in range(rank)]).contiguous().reshape(-1)
    tB = torch.empty_like(tA)
    A, B = cm(tA, shape), cm(tB, oshape)
    for xp, go in ((1, -1), (0, 0), (0, 1)):
        S.set_option("tiled_xpose", xp)
        S.set_option("tiled_gorder", go)
        try:
            tB.zero_()
            p = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(perm)))
            assert "family=tiled" in p.describe() and "d0:128" in p.describe(), p.describe()
            p.execute(cur())
            torch.cuda.synchronize()
            assert torch.equal(tB, want), (xp, go, p.describe())
        finally:
            S.set_option("tiled_xpose", 1)
            S.set_option("tiled_gorder", -1)
    # a scaled copy through the same path (a functor with a constant)
    if dt == "float64":
        p = S.make_plan(lambda x: 3 * x, None, None, B.size, (B, A.permutedims(perm)))
        p.execute(cur())
        torch.cuda.synchronize()
        assert torch.equal(tB, 3 * want)


@pytest.mark.parametrize("shape,perm,dt", [((257, 129, 65), (0, 2, 1), "float64"), ((17, 33, 65, 31), (0, 2, 1, 3), "float64"),
                                           ((999, 77), (0, 1), "float32"), ((1001, 5, 9), (0, 2, 1), "float32"),
                                           ((35, 64, 7), (0, 2, 1), "complex64"), ((131, 40, 3), (0, 2, 1), "int32")])
def test_rows_that_are_not_whole_aligned_vectors_move_as_vectors_plus_a_tail(shape, perm, dt):
    """STREAM, round 5: odd row lengths / rows starting at odd element offsets use 16-byte accesses at element alignment and one
    partial vector per row; compared with the scalar form (option stream_ua = 0) and with NumPy; also an n-ary map, a view that starts
    inside a vector, and a broadcast operand."""
    import torch
    rng = np.random.default_rng(11)
    npdt = np.dtype(dt)
    if npdt.kind == "c":
        a = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(npdt)
    elif npdt.kind == "i":
        a = rng.integers(-1000, 1000, shape).astype(npdt)
    else:
        a = rng.standard_normal(shape).astype(npdt)
    A = dview(a)
    want = np.transpose(a, perm)
    for ua in (1, 0):
        S.set_option("stream_ua", ua)
        try:
            B = dview(np.zeros(want.shape, dtype=npdt))
            p = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims(perm)))
            d = p.describe()
            assert "family=stream" in d, d
            vmax = 16 // npdt.itemsize
            if ua and npdt.itemsize in (4, 8) and shape[0] >= 4 * vmax and (shape[0] % vmax <= 1 or shape[0] >= 32 * vmax) and npdt.kind != "c":
                assert "element-aligned+tail" in d, d
            p.execute(cur())
            torch.cuda.synchronize()
            assert np.array_equal(host(B), want), (ua, d)
        finally:
            S.set_option("stream_ua", 1)
    if npdt.kind == "f":
        # n-ary map with a broadcast operand (stride 0 along dim 0) and a view that starts one element into the parent
        b = rng.standard_normal(want.shape).astype(npdt)
        col = rng.standard_normal((1,) + want.shape[1:]).astype(npdt)
        Bv, Cv = dview(b), dview(col)
        out = dview(np.zeros(want.shape, dtype=npdt))
        colb = S.StridedView(Cv.parent, want.shape, (0,) + Cv.strides[1:], 0)
        p = S.make_plan(lambda x, y, z: x * 2 + y - z, None, None, out.size, (out, A.permutedims(perm), Bv, colb))
        p.execute(cur())
        torch.cuda.synchronize()
        assert np.array_equal(host(out), want * 2 + b - col), p.describe()
        # a sub-view that drops the first and the last element of every row: starts inside a vector, odd or even length
        n0 = want.shape[0]
        big = dview(np.zeros(want.shape, dtype=npdt))
        sub_out = S.StridedView(big.parent, (n0 - 2,) + want.shape[1:], big.strides, 1)
        sub_in = S.StridedView(Bv.parent, (n0 - 2,) + want.shape[1:], Bv.strides, 1)
        p = S.make_plan(lambda x: x + 1, None, None, sub_out.size, (sub_out, sub_in))
        p.execute(cur())
        torch.cuda.synchronize()
        got = host(big)
        assert np.array_equal(got[1:-1], b[1:-1] + 1) and np.all(got[0] == 0) and np.all(got[-1] == 0), p.describe()
