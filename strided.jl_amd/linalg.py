"""Linear-algebra fronts -- mirror of /root/reference/src/linalg.jl.

The scalar / BLAS-1 fronts (src/linalg.jl:2-42) are broadcast callers and land on the natively
compiled functors (scale, axpy, axpby).  Matrix multiplication: the reference routes BLAS
floats to `gemm!` -- a dense contraction, out of scope for this path -- and
everything else to the generic `__mul!` (src/linalg.jl:130-162), which is a 3-operand
map-reduce with `initop`; that form is implemented here for every device element type.
"""
from __future__ import annotations

from .mapreduce import _mapreducedim_, copy_
from .stridedview import DimensionMismatch, StridedView


def rmul_(dst: StridedView, alpha) -> StridedView:
    """`rmul!(dst, alpha)` = mul!(dst, dst, alpha) (src/linalg.jl:2)."""
    return mul_(dst, dst, alpha)


def lmul_(alpha, dst: StridedView) -> StridedView:
    """`lmul!(alpha, dst)` = mul!(dst, alpha, dst) (src/linalg.jl:3)."""
    return mul_(dst, alpha, dst)


def mul_(dst: StridedView, a, b, alpha=1, beta=0) -> StridedView:
    """`mul!`: scalar forms (src/linalg.jl:5-22) or matrix product C = alpha*A*B + beta*C."""
    if isinstance(a, StridedView) and isinstance(b, StridedView):
        return _matmul_(dst, a, b, alpha, beta)
    if isinstance(b, StridedView):  # mul!(dst, alpha, src)
        alpha_, src = a, b
        if alpha_ == 1:
            return copy_(dst, src)
        return dst.assign(alpha_ * src)
    src, alpha_ = a, b  # mul!(dst, src, alpha)
    if alpha_ == 1:
        return copy_(dst, src)
    return dst.assign(src * alpha_)


def axpy_(a, X: StridedView, Y: StridedView) -> StridedView:
    """`axpy!(a, X, Y)`: Y .= a .* X .+ Y (src/linalg.jl:23-31)."""
    if a == 1:
        return Y.assign(X + Y)
    return Y.assign(a * X + Y)


def axpby_(a, X: StridedView, b, Y: StridedView) -> StridedView:
    """`axpby!(a, X, b, Y)` (src/linalg.jl:32-42)."""
    if b == 1:
        return axpy_(a, X, Y)
    if b == 0:
        return mul_(Y, a, X)
    return Y.assign(a * X + b * Y)


def _matmul_(C: StridedView, A: StridedView, B: StridedView, alpha=1, beta=0) -> StridedView:
    """`__mul!` (src/linalg.jl:130-162): C2=(m,n,1), A2=(m,1,k), B2=(1,n,k) views and one
    `_mapreducedim!(f, +, initop, (m,n,k), (C2, A2, B2))`."""
    if C.ndim != 2 or A.ndim != 2 or B.ndim != 2:
        raise DimensionMismatch("mul_ needs matrices")
    if not (C.size[0] == A.size[0] and C.size[1] == B.size[1] and A.size[1] == B.size[0]):
        raise DimensionMismatch(f"A has size {A.size}, B has size {B.size}, C has size {C.size}")
    m, n = C.size
    k = A.size[1]
    A2 = StridedView(A.parent, (m, 1, k), (A.strides[0], 0, A.strides[1]), A.offset, A.op)
    Bt = B.permutedims((1, 0))
    B2 = StridedView(B.parent, (1, n, k), (0, Bt.strides[0], Bt.strides[1]), B.offset, B.op)
    C2 = StridedView(C.parent, (m, n, 1), (C.strides[0], C.strides[1], 0), C.offset, C.op)
    if alpha == 0 or k == 0:
        return rmul_(C, beta)
    if beta == 0:
        initop = "zero"
    elif beta == 1:
        initop = None
    else:
        initop = ("scale", beta)
    if alpha == 1:
        f = lambda x, y: x * y  # noqa: E731
    else:
        f = lambda x, y: x * y * alpha  # noqa: E731
    _mapreducedim_(f, "+", initop, (m, n, k), (C2, A2, B2))
    return C
