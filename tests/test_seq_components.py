"""CPU: the dependency analysis behind recorded sequences and eager direct dispatch (csrc/smr_seq.cpp: components_of; footprints =
bounding byte ranges of every operand, smr_api.cpp: footprint) -- pure host arithmetic, so it runs without a device.  Two executions belong
to one component when one writes bytes the other reads or writes; a component keeps its recorded order on one hardware queue, different
components run concurrently (the device form of /root/reference/src/mapreduce.jl:203-223: spawn what is independent, wait where it must).
The analysis may only err on the side of MORE ordering: every pair that really conflicts must share a component."""
import numpy as np

import strided_jl_amd as S


def view(a):
    return S.StridedView(a)


def seq_of(*plans):
    q = S.Sequence()
    for p in plans:
        q.add(p)
    return q


def copy_plan(dst, src):
    return S.make_plan(lambda x: x, None, None, dst.size, (dst, src))


def test_bench_step_has_two_components():
    n = 16
    a = np.zeros((n,) * 4, order="F")
    A, B, C = view(a), view(np.zeros_like(a)), view(np.zeros_like(a))
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    p2 = copy_plan(B, A.permutedims((3, 2, 1, 0)))
    p3 = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in perms))
    assert seq_of(p2, p3).components() == [0, 1]          # both only READ A
    assert seq_of(p2, p3, p2, p3).components() == [0, 1, 0, 1]


def test_read_after_write_write_after_read_write_after_write():
    a = np.zeros((40, 30), order="F")
    A, B, C, D = (view(np.zeros_like(a)) for _ in range(4))
    raw = seq_of(copy_plan(B, A), copy_plan(C, B))              # the second reads what the first wrote
    war = seq_of(copy_plan(B, A), copy_plan(A, C))              # the second overwrites what the first reads
    waw = seq_of(copy_plan(B, A), copy_plan(B, C))              # both write B
    ind = seq_of(copy_plan(B, A), copy_plan(D, C))              # nothing in common
    assert raw.components() == [0, 0] and war.components() == [0, 0] and waw.components() == [0, 0]
    assert ind.components() == [0, 1]


def test_a_join_merges_its_producers():
    a = np.zeros((20, 20, 20), order="F")
    A, X, Y, Z, W = (view(np.zeros_like(a)) for _ in range(5))
    px = S.make_plan(lambda x: x * 3, None, None, X.size, (X, A.permutedims((2, 1, 0))))
    py = S.make_plan(lambda x: x - 1, None, None, Y.size, (Y, A))
    pw = copy_plan(W, A.permutedims((1, 0, 2)))               # independent of the other three
    pz = S.make_plan(lambda x, y: x * y, None, None, Z.size, (Z, X, Y))
    assert seq_of(px, py, pw, pz).components() == [0, 0, 1, 0]  # the consumer ties X's and Y's producers together; W stays apart
    assert seq_of(px, py, pw).components() == [0, 1, 2]


def test_sub_views_of_one_parent():
    a = np.zeros((64, 64), order="F")
    P, Q = view(a), view(np.zeros_like(a))
    left, right = P.sview(slice(None), slice(0, 32)), P.sview(slice(None), slice(32, 64))    # disjoint column blocks: disjoint byte ranges
    qa, qb = Q.sview(slice(None), slice(0, 32)), Q.sview(slice(None), slice(32, 64))
    assert seq_of(copy_plan(left, qa), copy_plan(right, qb)).components() == [0, 1]
    mid = P.sview(slice(None), slice(16, 48))
    assert seq_of(copy_plan(left, qa), copy_plan(mid, qb)).components() == [0, 0]             # columns 16..31 are written by both
    # reversed views: the footprint is the range the view really covers, whatever the sign of its strides
    rright = P.sview(slice(None), slice(63, 31, -1))
    assert seq_of(copy_plan(left, qa), copy_plan(rright, qb)).components() == [0, 1]
    # row blocks interleave in memory (column-major): their bounding ranges overlap although no element is shared -- the analysis is
    # conservative and orders them
    top, bottom = P.sview(slice(0, 32), slice(None)), P.sview(slice(32, 64), slice(None))
    qt, qbm = Q.sview(slice(0, 32), slice(None)), Q.sview(slice(32, 64), slice(None))
    assert seq_of(copy_plan(top, qt), copy_plan(bottom, qbm)).components() == [0, 0]


def test_reductions_read_their_destination():
    a = np.zeros((30, 40), order="F")
    A = view(a)
    r = np.zeros((30, 1), order="F")
    R, T = view(r), view(np.zeros_like(r))
    from strided_jl_amd.broadcast import promoteshape
    acc = S.make_plan(lambda x: x, "+", None, A.size, promoteshape(A.size, R, A))               # R = R + sum(A, dims=2): reads R
    use = copy_plan(T, R)
    other = S.make_plan(lambda x: x, "+", None, A.size, promoteshape(A.size, T, A))
    assert seq_of(acc, use).components() == [0, 0]
    assert seq_of(acc, acc).components() == [0, 0]
    assert seq_of(acc, other).components() == [0, 1]


def test_rebound_base_pointers_decide():
    n = 12
    a = np.zeros((n, n, n), order="F")
    A = view(a)
    pool = np.zeros((4, n ** 3))
    B0 = S.StridedView(pool[0], (n, n, n), (1, n, n * n), 0)
    p = copy_plan(B0, A.permutedims((2, 0, 1)))
    q = S.Sequence()
    for i in (0, 1, 2, 1):
        q.add(p, bases=[pool[i].ctypes.data, a.ctypes.data])
    assert q.components() == [0, 1, 2, 1]


def test_which_packets_acquire_and_when_launches_may_be_self_released():
    """Round 5 (csrc/smr_seq.cpp): inside a replay only executions that READ what the sequence WRITES acquire; launches are
    self-released (write-through stores, no release fence) only while the whole footprint fits half the Infinity Cache."""
    n = 16
    a = np.zeros((n,) * 4, order="F")
    A, B, C = view(a), view(np.zeros_like(a)), view(np.zeros_like(a))
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    p2 = copy_plan(B, A.permutedims((3, 2, 1, 0)))
    p3 = S.make_plan(lambda w, x, y, z: w + x + y + z, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in perms))
    acq, fp, resident = seq_of(p2, p3).fences()
    assert acq == [0, 0] and resident                      # the bench step: nobody writes A
    assert fp == 3 * a.nbytes                              # A, B, C once each (A's four views are one range)
    m = np.zeros((40, 30), order="F")
    X, Y, Z, ACC = (view(np.zeros_like(m)) for _ in range(4))
    r1 = copy_plan(Y, X)                                   # reads X only
    r2 = S.make_plan(lambda u, v: u + v, None, None, Z.size, (Z, Y, X))   # reads what r1 wrote
    r3 = S.make_plan(lambda u, v: u + v, None, None, ACC.size, (ACC, ACC, Z))   # in place: reads its own destination
    red = S.make_plan(lambda u: u, "+", None, X.size, (S.StridedView(np.zeros(1), X.size, (0, 0), 0), X))
    assert seq_of(r1, r2, r3).fences()[0] == [0, 1, 1]
    assert seq_of(r1, red).fences()[0] == [0, 1]           # a reduction accumulates into (reads) its destination
    assert seq_of(r2, r1).fences()[0] == [1, 0]            # order in the list does not matter: the next replay's r2 follows this one's r1
    big = np.zeros((4096, 4096), order="F")                # 128 MiB each: two of them exceed the 128 MiB default
    P, Q = view(big), view(np.zeros_like(big))
    acq, fp, resident = seq_of(copy_plan(Q, P.permutedims((1, 0)))).fences()
    assert acq == [0] and fp == 2 * big.nbytes and not resident
    S.set_option("self_release_max_total", 1 << 40)
    try:
        assert seq_of(copy_plan(Q, P.permutedims((1, 0)))).fences()[2]
    finally:
        S.set_option("self_release_max_total", 128 << 20)
