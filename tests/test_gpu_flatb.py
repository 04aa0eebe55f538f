"""GPU: the batched form of the FLAT family (round 4; csrc/smr_k_flat.hip: flatb_body) -- unary maps whose first dims form one
contiguous block on both sides, in different orders, with the batch dim right behind: batched transposes of small matrices /
tensors ((9,11,N) -> (11,9,N); TensorOperations-style `tensoradd!` over /root/reference/src/broadcast.jl:27-37).  Bit-exact vs NumPy."""
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


CASES = [((9, 11, 3000), (1, 0, 2)), ((5, 9, 4001), (1, 0, 2)), ((17, 23, 700), (1, 0, 2)), ((3, 4, 5, 2000), (2, 0, 1, 3)), ((3, 4, 5, 2000), (1, 2, 0, 3)),
         ((7, 6, 333, 9), (1, 0, 2, 3)), ((2, 2, 2, 2, 5000), (3, 1, 2, 0, 4)), ((16, 16, 999), (1, 0, 2)), ((9, 11, 1), (1, 0, 2)), ((9, 11, 2, 3001), (1, 0, 2, 3)),
         ((9, 11, 70, 60), (1, 0, 3, 2)), ((4, 8, 60, 50, 7), (1, 0, 4, 2, 3)), ((13, 5, 40, 130), (1, 0, 3, 2))]


@pytest.mark.parametrize("shape,perm", CASES)
@pytest.mark.parametrize("dt", [np.float64, np.float32, np.complex64, np.complex128, np.int32])
def test_batched_blocks(shape, perm, dt):
    import torch
    rng = np.random.default_rng(sum(shape))
    if np.issubdtype(dt, np.integer):
        a = rng.integers(-1000, 1000, size=shape).astype(dt)
    elif np.issubdtype(dt, np.complexfloating):
        a = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
    else:
        a = rng.standard_normal(shape).astype(dt)
    A = dview(a)
    out = dview(np.zeros(tuple(shape[i] for i in perm), dtype=dt))
    plan = S.make_plan(lambda x: x, None, None, out.size, (out, A.permutedims(perm)))
    if shape[:3] in ((9, 11, 3000), (5, 9, 4001)) or (shape[:2] == (9, 11) and len(shape) == 4 and np.dtype(dt).itemsize >= 8):
        assert "batched" in plan.describe(), plan.describe()
    plan.execute()
    torch.cuda.synchronize()
    assert np.array_equal(out.toarray(), np.transpose(a, perm))
    # with a function and a conjugated input view, through the one-shot entry point (runtime-compiled functor)
    if not np.issubdtype(dt, np.integer):
        v = A.permutedims(perm)
        if np.issubdtype(dt, np.complexfloating):
            v = v.conj() if hasattr(v, "conj") else v
        S.map_(lambda x: x * 2 - 1, out, v)
        torch.cuda.synchronize()
        want = np.transpose(a, perm)
        if np.issubdtype(dt, np.complexfloating) and hasattr(A, "conj"):
            want = np.conj(want)
        assert np.array_equal(out.toarray(), want * dt(2) - dt(1))


def test_batched_form_respects_offsets_and_outer_strides():
    """blocks inside a bigger parent: the batch dims keep the parent's strides (a sub-range of the batches), offsets are odd"""
    import torch
    rng = np.random.default_rng(7)
    a = rng.standard_normal((9, 11, 2000, 4))
    A = dview(a)
    sub = A.sview(slice(None), slice(None), slice(3, 1503), slice(1, 3))          # 1500 batches x 2 outer
    out_full = dview(np.zeros((11, 9, 2000, 4)))
    out = out_full.sview(slice(None), slice(None), slice(3, 1503), slice(1, 3))
    plan = S.make_plan(lambda x: x, None, None, out.size, (out, sub.permutedims((1, 0, 2, 3))))
    assert "batched" in plan.describe(), plan.describe()
    plan.execute()
    torch.cuda.synchronize()
    want = np.zeros((11, 9, 2000, 4))
    want[:, :, 3:1503, 1:3] = np.transpose(a[:, :, 3:1503, 1:3], (1, 0, 2, 3))
    assert np.array_equal(out_full.toarray(), want)
