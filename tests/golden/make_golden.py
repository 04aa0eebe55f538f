#!/usr/bin/env python3
"""Generates tests/golden/strided_golden.npz: inputs and expected outputs of the reference's
headline operations, computed with plain NumPy as an independent ground truth.

The reference (Strided.jl) cannot run here (no Julia runtime) and its own test-suite stores no
vectors -- every assertion is `Base-Julia(x) == Strided(x)` evaluated at run time.  These
fixtures restate the Base-Julia side of those assertions (test/othertests.jl:1-128,253-333 and
README.md:56-105) with NumPy, whose elementwise semantics for + - * / conj transpose sum are the
same IEEE operations.  Column-major ("F") arrays throughout, like Julia.

    python tests/golden/make_golden.py        # rewrites the .npz (deterministic)
"""
import os

import numpy as np

rng = np.random.default_rng(20240927)
G = {}


def F(a):
    return np.asfortranarray(a)


# permutedims! on index-valued data (A[i] = linear index: exact, transpose-detecting)
A = F(np.arange(7 * 5 * 6 * 4, dtype=np.float64).reshape((7, 5, 6, 4), order="F"))
G["perm_in"] = A
for name, p in (("4321", (3, 2, 1, 0)), ("2341", (1, 2, 3, 0)), ("3412", (2, 3, 0, 1))):
    G[f"perm_{name}"] = F(A.transpose(p))
# README.md:72,83  symmetrise and scaled transpose
S_ = F(rng.standard_normal((33, 33)))
G["sym_in"] = S_
G["sym_out"] = F((S_ + S_.T) / 2)
G["scaledT_out"] = F(3 * S_.T)
# README.md:104  4-way permuted sum, left-to-right association
C4 = F(rng.standard_normal((6, 6, 6, 6)))
G["sum4_in"] = C4
G["sum4_out"] = F(((C4 + C4.transpose(1, 2, 3, 0)) + C4.transpose(2, 3, 0, 1)) + C4.transpose(3, 0, 1, 2))
# conj! / adjoint!  (test/othertests.jl:10-11)
Z = F(rng.standard_normal((5, 7)) + 1j * rng.standard_normal((5, 7)))
G["cplx_in"] = Z
G["conj_out"] = F(np.conj(Z))
G["adjoint_out"] = F(np.conj(Z.T))
# partial reductions with initop (test/othertests.jl:71-102)
R = F(rng.random((4, 5, 6)) + 1j * rng.random((4, 5, 6)))
O = F(rng.random((4, 1, 6)) + 1j * rng.random((4, 1, 6)))
beta = 0.3 - 0.7j
red = np.sin(R).sum(axis=1, keepdims=True)
G["red_in"], G["red_dest"], G["red_beta"] = R, O, np.array([beta])
G["red_none"] = F(red + O)            # initop = nothing / identity
G["red_zero"] = F(red)                # x -> 0
G["red_scale"] = F(red + beta * O)    # x -> beta*x
G["red_const"] = F(red + beta)        # x -> beta
G["red_conj"] = F(red + np.conj(O))   # conj
# complete reductions (test/othertests.jl:113-116,125)
X = F(rng.standard_normal((10, 9, 8)))
G["full_in"] = X
G["full_sum"] = np.array([X.sum()])
G["full_maxabs"] = np.array([np.abs(X).max()])
G["full_min"] = np.array([X.min()])
G["full_count_neg"] = np.array([(X < 0).sum()], dtype=np.int64)
G["full_abs2"] = np.array([(X * X).sum()])
P = F(rng.random((5, 5, 5)))
G["prod_in"] = P
G["prod_exp"] = np.array([np.exp(P.sum())])
# README.md:89 compute-bound map in Float32: float64 truth rounded once
E32 = F(rng.random((64, 48)).astype(np.float32))
e64 = E32.astype(np.float64)
G["expr_in"] = E32
G["expr_out"] = F((e64 * np.exp(-2 * e64) + np.sin(e64 * e64)).astype(np.float32))
G["expr_scale"] = F((np.abs(e64 * np.exp(-2 * e64)) + np.abs(np.sin(e64 * e64))).astype(np.float32))
# generic matmul on integer-valued complex data (test/othertests.jl:253-296): exact
M1 = F(rng.integers(-100, 101, (9, 9)) + 1j * rng.integers(-100, 101, (9, 9)))
M2 = F(rng.integers(-100, 101, (9, 9)) + 1j * rng.integers(-100, 101, (9, 9)))
M3 = F(rng.integers(-100, 101, (9, 9)) + 1j * rng.integers(-100, 101, (9, 9)))
al, be = 2 + 1j, 3 - 1j
G["mm_a"], G["mm_b"], G["mm_c"] = M1, M2, M3
G["mm_alpha_beta"] = np.array([al, be])
G["mm_out"] = F(be * M3 + al * (M1.conj().T @ M2.T))          # C = beta*C + alpha*A'*transpose(B)
G["mm_out_conjdest"] = F(np.conj(be) * M3 + np.conj(al * (M1 @ M2)))  # mul!(conj(C), A, B, alpha, beta)

# ---- round 6: every documented deviation of DESIGN.md section 4 and every BASELINE config at reduced size, so that
# ONE run of make_golden.jl settles them all.  A second generator: the arrays above keep their values.
rng6 = np.random.default_rng(20261001)
# (a) src/mapreduce.jl:409 -- reversed destination + initop on a working set above 32 KiB, laid out so that the planner CUTS the kept
#     (reversed) dim into blocks (the layout family tools/fuzz_more.py found in round 5; tests/test_golden.py asserts the cut).  The
#     oracle and the device read the line as `!= 0`; read literally (`> 0`) every block after the first keeps its stale
#     destination values.  out[k] = initop(out[k]) + sum_{i,j} (x + y + z)[i, j, k], destination seen through a reversed range.
#     Inputs are NOT stored: element i of a parent is mod(i * a, m) / m + 0.25 (exact in any IEEE language) -- REV409 below.
REV409 = dict(dims=(100, 8, 159), dest=((0, 0, -1), 158, 159, (104729, 1013)),
              ins=[((160, 0, 1), 0, 15999, (7919, 1009)), ((-1280, -160, 1), 127840, 127999, (7920, 1013)), ((3, 2400, 300), 0, 64498, (7921, 1019))])


def formula(n, a, m):
    i = np.arange(n, dtype=np.int64)
    return ((i * a) % m).astype(np.float64) / m + 0.25


def strided(par, dims, st, off):
    idx = off + sum(s_ * ix for s_, ix in zip(st, np.meshgrid(*(np.arange(d) for d in dims), indexing="ij")))
    return par[idx]


_d = REV409["dims"]
for _st, _off, _n, _am in REV409["ins"]:
    _lo = _off + sum(min(0, (d - 1) * s_) for d, s_ in zip(_d, _st))
    _hi = _off + sum(max(0, (d - 1) * s_) for d, s_ in zip(_d, _st))
    assert _lo >= 0 and _hi < _n, (_st, _lo, _hi, _n)
tot9 = None
for _st, _off, _n, _am in REV409["ins"]:
    _v = strided(formula(_n, *_am), _d, _st, _off)
    tot9 = _v if tot9 is None else tot9 + _v
red9 = tot9.sum(axis=(0, 1))[::-1]            # element p of the PARENT vector holds out[k] for k = 158 - p
O9 = formula(159, *REV409["dest"][3])
beta9 = 0.375
G["rev409_beta"] = np.array([beta9])
for key, init in (("none", O9), ("zero", 0 * O9), ("scale", beta9 * O9), ("const", beta9 + 0 * O9), ("conj", np.conj(O9))):
    G[f"rev409_{key}"] = init + red9
# (b) the `_computeblocks` termination case: negative strides make every block weight <= 0 and the reference's halving loops
#     (src/mapreduce.jl:491-498) are suspected never to end; the oracle breaks out.  dims (66, 2, 19), a 3-input map.
gd = (66, 2, 19)
gviews = [((38, 19, -1), 18), ((38, 2, 4), 0), ((-1, 66, 0), 65), ((0, 1, 0), 0)]   # (strides, offset): destination first
gsizes = [2508, 2545, 132, 2]
gpar = [np.zeros(gsizes[0])] + [rng6.standard_normal(n) for n in gsizes[1:]]


def gview(par, st, off):
    return np.lib.stride_tricks.as_strided(par[off:], shape=gd, strides=tuple(8 * x for x in st), writeable=False)


gsum = (gview(gpar[1], *gviews[1]) + gview(gpar[2], *gviews[2])) + gview(gpar[3], *gviews[3])
gout = gpar[0].copy()
i0, i1, i2 = np.meshgrid(*(np.arange(d) for d in gd), indexing="ij")
gout[gviews[0][1] + 38 * i0 + 19 * i1 - i2] = gsum
G["guard_in1"], G["guard_in2"], G["guard_in3"], G["guard_out"] = gpar[1], gpar[2], gpar[3], gout
# (c) stride-0 broadcast operands (src/broadcast.jl:41-65): a column, a row and a scalar
G["bc0_col"], G["bc0_row"] = F(rng6.standard_normal((5, 1))), F(rng6.standard_normal((1, 6)))
G["bc0_out"] = F(G["bc0_col"] + G["bc0_row"] * 2.5)
# (d) an offset, stepped sub-view seen through a permutation (test/othertests.jl:134-189)
SV = F(rng6.standard_normal((12, 10)))
G["sv_in"] = SV
G["sv_out"] = F(2 * SV[2:9, 1:10:2].T)       # Julia: 2 .* permutedims(sview(A, 3:9, 2:2:10), (2, 1))
# (e) BASELINE configs at reduced size: configs[0] = sym_*, [1] = perm_4321, [2] = sum4_*, [4] = expr_* above; configs[3] here
C4 = F((rng6.random((16, 16, 8)) * 2 - 1).astype(np.float32))
G["c4_in"] = C4
G["c4_abs2"] = np.array([(C4.astype(np.float64) ** 2).sum()])   # float64 truth; the engine accumulates in Float32 (rtol 1e-5)

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "strided_golden.npz")
np.savez_compressed(out, **G)
print("wrote", out, os.path.getsize(out), "bytes,", len(G), "arrays")
