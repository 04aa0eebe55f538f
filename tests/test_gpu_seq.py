"""GPU: recorded sequences (smr_seq, csrc/smr_seq.cpp) -- the library's own replay of a list of plan executions as AQL
packets on its HSA queues, one queue per dependency component.  The contract under test: whatever runs concurrently, the
results are those of executing the recorded list in order (the device form of src/mapreduce.jl:203-223: spawn what is
independent, wait where it must).  Every case is compared bit for bit with NumPy and with the same plans executed one by
one on a stream."""
import numpy as np
import pytest

import strided_jl_amd as S

pytestmark = pytest.mark.gpu


def dview(arr):
    import torch
    a = np.asfortranarray(arr)
    t = torch.from_numpy(a.ravel(order="F").copy()).cuda()
    st, s = [], 1
    for d in a.shape:
        st.append(s)
        s *= d
    return S.StridedView(t, a.shape, tuple(st), 0)


def stream():
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def sync():
    import torch
    torch.cuda.synchronize()


def field(info, key):
    for tok in info.split():
        if tok.startswith(key + "="):
            return tok.split("=", 1)[1]
    raise KeyError(key + " not in: " + info)


PERMS = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]


@pytest.mark.parametrize("n", [8, 16, 32])
@pytest.mark.parametrize("queues", [1, 4])
def test_bench_step_overlapped_equals_in_order(n, queues):
    """The bench step: permutedims!(B, A, (4,3,2,1)) and C .= sum of 4 permuted views of A -- two components."""
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n,) * 4)
    A, B, C = dview(a), dview(np.zeros_like(a)), dview(np.zeros_like(a))
    p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    p3 = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in PERMS))
    q = S.Sequence().add(p2).add(p3)
    q.set("queues", queues)
    q.set("slices", 1)   # (round 5 cuts the heavier chain in two by default: tests/test_gpu_round5.py)
    q.run(3, stream())
    q.wait()
    sync()
    info = q.info()
    assert field(info, "backend") == "aql", info
    assert field(info, "components") == "2" and field(info, "queues") == str(min(queues, 2)), info
    assert np.array_equal(B.toarray(), np.transpose(a, (3, 2, 1, 0)))
    want = ((np.transpose(a, PERMS[0]) + np.transpose(a, PERMS[1])) + np.transpose(a, PERMS[2])) + np.transpose(a, PERMS[3])
    assert np.array_equal(C.toarray(), want)


def test_dependent_chain_is_one_component():
    """B = A', C = B + B' (reads what the first launch wrote), D = 2C - B: read-after-write keeps everything on one queue."""
    rng = np.random.default_rng(5)
    a = rng.standard_normal((96, 96))
    A, B, C, D = dview(a), dview(np.zeros_like(a)), dview(np.zeros_like(a)), dview(np.zeros_like(a))
    p1 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((1, 0))))
    p2 = S.make_plan(lambda x, y: x + y, None, None, A.size, (C, B, B.permutedims((1, 0))))
    p3 = S.make_plan(lambda x, y: 2 * x - y, None, None, A.size, (D, C, B))
    q = S.Sequence().add(p1).add(p2).add(p3)
    q.run(2, stream())
    q.wait()
    sync()
    info = q.info()
    assert field(info, "components") == "1" and field(info, "queues") == "1", info
    b = a.T
    c = b + b.T
    assert np.array_equal(B.toarray(), b)
    assert np.array_equal(C.toarray(), c)
    assert np.array_equal(D.toarray(), 2 * c - b)


def test_two_chains_and_a_join():
    """Two independent producers and one consumer of both: the consumer ties them into one component (no cross-queue edge
    exists by construction), the result is the in-order one."""
    rng = np.random.default_rng(6)
    a = rng.standard_normal((40, 50, 30))
    A = dview(a)
    X, Y, Z = dview(np.zeros((30, 50, 40))), dview(np.zeros((30, 50, 40))), dview(np.zeros((30, 50, 40)))
    p1 = S.make_plan(lambda x: x * 3, None, None, X.size, (X, A.permutedims((2, 1, 0))))
    p2 = S.make_plan(lambda x: x - 1, None, None, Y.size, (Y, A.permutedims((2, 1, 0))))
    p3 = S.make_plan(lambda x, y: x * y, None, None, Z.size, (Z, X, Y))
    q = S.Sequence().add(p1).add(p2).add(p3)
    q.run(1, stream())
    q.wait()
    sync()
    assert field(q.info(), "components") == "1", q.info()
    at = np.transpose(a, (2, 1, 0))
    assert np.array_equal(Z.toarray(), (at * 3) * (at - 1))


def test_replays_are_ordered_among_themselves():
    """An accumulating execution (y .= y + a*x) replayed r times must see its own previous result r times; next to it an
    independent chain on another queue."""
    rng = np.random.default_rng(7)
    x = rng.integers(-8, 8, (64, 64, 16)).astype(np.float64)
    y0 = rng.integers(-8, 8, (64, 64, 16)).astype(np.float64)
    Xv, Yv = dview(x), dview(y0)
    Wv, Vv = dview(np.zeros((16, 64, 64))), dview(x)
    pa = S.make_plan(lambda y, xx: y + 2 * xx, None, None, Yv.size, (Yv, Yv, Xv))
    pb = S.make_plan(lambda v: v, None, None, Wv.size, (Wv, Vv.permutedims((2, 1, 0))))
    q = S.Sequence().add(pa).add(pb)
    reps = 37
    q.run(reps, stream())
    q.wait()
    sync()
    assert field(q.info(), "components") == "2", q.info()
    assert np.array_equal(Yv.toarray(), y0 + 2 * reps * x)  # small integers: exact in Float64
    assert np.array_equal(Wv.toarray(), np.transpose(x, (2, 1, 0)))


def test_rebound_bases_rotate_over_buffers():
    """One plan recorded four times with different base pointers (four disjoint output buffers): four components."""
    import torch
    rng = np.random.default_rng(8)
    n = 24
    a = rng.standard_normal((n, n, n))
    A = dview(a)
    pool = torch.zeros(4, n ** 3, dtype=torch.float64, device="cuda")
    B0 = S.StridedView(pool[0], (n, n, n), (1, n, n * n), 0)
    p = S.make_plan(lambda v: v, None, None, B0.size, (B0, A.permutedims((2, 0, 1))))
    q = S.Sequence()
    a_ptr = A.parent.data_ptr()
    for i in range(4):
        q.add(p, bases=[pool[i].data_ptr(), a_ptr])
    q.run(2, stream())
    q.wait()
    sync()
    info = q.info()
    assert field(info, "components") == "4" and field(info, "queues") == "4", info
    want = np.transpose(a, (2, 0, 1)).ravel(order="F")
    for i in range(4):
        assert np.array_equal(pool[i].cpu().numpy(), want), i


def test_reductions_with_partials_inside_a_sequence():
    """Executions with several launches (partials + folding pass) keep their internal order; the plan's scratch belongs to
    its footprint, so the same plan recorded twice is one component."""
    rng = np.random.default_rng(9)
    a = rng.integers(-4, 5, (300, 200, 90)).astype(np.float64)
    A = dview(a)
    R1, R2 = dview(np.zeros((1, 1, 90))), dview(np.zeros((300, 1, 1)))
    from strided_jl_amd.broadcast import promoteshape
    p1 = S.make_plan(lambda v: v, "+", None, A.size, promoteshape(A.size, R1, A))
    p2 = S.make_plan(S.fn.abs2, "+", None, A.size, promoteshape(A.size, R2, A))
    q = S.Sequence().add(p1).add(p2)
    q.run(1, stream())
    q.wait()
    sync()
    # mapreducedim!-style accumulation into a zeroed destination, integers: exact
    assert np.array_equal(R1.toarray().ravel(), a.sum(axis=(0, 1)))
    assert np.array_equal(R2.toarray().ravel(), (a * a).sum(axis=(1, 2)))


def test_runtime_compiled_kernel_falls_back_to_hip_in_order():
    rng = np.random.default_rng(10)
    a = rng.standard_normal((64, 64))
    A, B, C = dview(a), dview(np.zeros_like(a)), dview(np.zeros_like(a))
    p1 = S.make_plan(lambda x, y: x * y - x / 3, None, None, A.size, (B, A, A.permutedims((1, 0))))
    if p1.jit_compile() <= 0:
        pytest.skip("the runtime compiler is unavailable on this box")
    p2 = S.make_plan(lambda x: x + 1, None, None, A.size, (C, B))
    q = S.Sequence().add(p1).add(p2)
    q.run(2, stream())
    q.wait()
    sync()
    info = q.info()
    want_b = a * a.T - a / 3
    assert np.allclose(B.toarray(), want_b, rtol=1e-15, atol=0)
    assert np.array_equal(C.toarray(), B.toarray() + 1)
    assert field(info, "backend") in ("aql", "hip"), info


def test_more_packets_than_the_ring_holds():
    """20,000 packets per queue through a 16,384-slot ring: the writer waits for the packet processor and wraps."""
    x = np.arange(64 * 4, dtype=np.float64).reshape(64, 4)
    Y, X = dview(np.zeros_like(x)), dview(x)
    Z = dview(np.zeros((4, 64)))
    pa = S.make_plan(lambda y, xx: y + xx, None, None, Y.size, (Y, Y, X))
    pb = S.make_plan(lambda v: v, None, None, Z.size, (Z, X.permutedims((1, 0))))
    q = S.Sequence().add(pa).add(pb)
    reps = 20000
    q.run(reps, stream())
    q.wait()
    sync()
    assert np.array_equal(Y.toarray(), reps * x)
    assert np.array_equal(Z.toarray(), x.T)


def test_stream_work_before_and_after_is_ordered():
    """Work queued on the caller's stream before smr_seq_run completes first; work queued afterwards sees the replay."""
    import torch
    rng = np.random.default_rng(11)
    a = rng.standard_normal((128, 128))
    A, B = dview(np.zeros_like(a)), dview(np.zeros_like(a))
    src = torch.from_numpy(a.ravel(order="F").copy()).cuda()
    p = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((1, 0))))
    q = S.Sequence().add(p)
    A.parent.copy_(src)          # queued on the current stream BEFORE the replay
    q.run(1, stream())
    out = B.parent.clone()       # queued AFTER the replay
    q.wait()
    sync()
    assert np.array_equal(out.cpu().numpy().reshape(128, 128, order="F"), a.T)


@pytest.mark.parametrize("n", [8, 16, 32, 48])
@pytest.mark.parametrize("slices", [2, 3, 4])
def test_single_launch_components_cut_into_block_ranges(n, slices):
    """A component that is ONE launch of independent workgroups is cut into contiguous block ranges, one hardware queue each (the
    device form of _mapreduce_threaded!, src/mapreduce.jl:195-227); the step must come out bit-identical, replay after replay."""
    rng = np.random.default_rng(100 + n)
    a = rng.standard_normal((n,) * 4)
    A, B, C = dview(a), dview(np.zeros_like(a)), dview(np.zeros_like(a))
    p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    p3 = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in PERMS))
    q = S.Sequence().add(p2).add(p3)
    q.set("queues", 8)            # (the default of 4 leaves room for 2 slices x 2 components only)
    q.set("slices", slices)
    q.run(5, stream())
    q.wait()
    sync()
    info = q.info()
    assert field(info, "backend") == "aql", info
    if n == 32:  # both launches are one-shot forms there (the persistent forms of bigger problems are not sliceable)
        assert field(info, "sliced") == "2" and field(info, "queues") == str(2 * slices), info
    assert np.array_equal(B.toarray(), np.transpose(a, (3, 2, 1, 0)))
    want = ((np.transpose(a, PERMS[0]) + np.transpose(a, PERMS[1])) + np.transpose(a, PERMS[2])) + np.transpose(a, PERMS[3])
    assert np.array_equal(C.toarray(), want)


def test_components_with_several_executions_are_not_sliced():
    rng = np.random.default_rng(77)
    a = rng.standard_normal((64, 64, 32))
    A, B, C = dview(a), dview(np.zeros((32, 64, 64))), dview(np.zeros((32, 64, 64)))
    p1 = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims((2, 1, 0))))
    p2 = S.make_plan(lambda x: x * 2, None, None, C.size, (C, B))                   # reads what p1 wrote: one component of two
    q = S.Sequence().add(p1).add(p2)
    q.set("slices", 4)
    q.run(3, stream())
    q.wait()
    sync()
    info = q.info()
    assert field(info, "components") == "1" and field(info, "sliced") == "0" and field(info, "queues") == "1", info
    assert np.array_equal(C.toarray(), np.transpose(a, (2, 1, 0)) * 2)
