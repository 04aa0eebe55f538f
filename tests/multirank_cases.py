"""Case table of the multi-process test of csrc/smr_comm.cpp (tests/test_gpu_multirank.py): every rank and the parent build
the SAME logical problems from it (inputs from a seeded generator)."""
import numpy as np

import strided_jl_amd as S

fn = S.fn

F = {
    "ident": lambda x: x,
    "abs2": fn.abs2,
    "mul": lambda x, y: x * y,
    "gt0": lambda x: x > 0,
    "scaled_diff": lambda x, y: 2 * x - y,
}


def cases(world):
    """name, f, op, initop, dims, dest (kept shape, dtype, initial value), inputs [(dtype, perm or None, local?)], exact?"""
    small = 7 if world == 8 else 3  # kept extents below the number of ranks force the split of a REDUCED dim + the all-reduce
    out = [
        # BASELINE config 4 at 1/8 size, block-partitioned input: every rank holds ONLY its slab of dim 3
        dict(name="c4_abs2_sum_f32", f="abs2", op="+", initop=None, dims=(4096, 512, 64), kept=(1, 1, 1), ddt=np.float32, dinit=0.0,
             ins=[(np.float32, None, True)], exact=False),
        # complete reductions: the five initop forms (test/othertests.jl:76-102, src/linalg.jl:146-160), destination content = 3
        dict(name="sum_init_none", f="ident", op="+", initop=None, dims=(40, 30, 16), kept=(1, 1, 1), ddt=np.float64, dinit=3.0,
             ins=[(np.float64, (2, 0, 1), False)], exact=False),
        dict(name="sum_init_zero", f="ident", op="+", initop="zero", dims=(40, 30, 16), kept=(1, 1, 1), ddt=np.float64, dinit=3.0,
             ins=[(np.float64, None, False)], exact=False),
        dict(name="sum_init_identity", f="abs2", op="+", initop="identity", dims=(40, 30, 16), kept=(1, 1, 1), ddt=np.float64, dinit=3.0,
             ins=[(np.float64, None, True)], exact=False),
        dict(name="sum_init_scale", f="mul", op="+", initop=("scale", 0.5), dims=(40, 30, 16), kept=(1, 1, 1), ddt=np.float64, dinit=3.0,
             ins=[(np.float64, None, False), (np.float64, (1, 0, 2), False)], exact=False),
        dict(name="sum_init_const", f="ident", op="+", initop=("const", -2.0), dims=(40, 30, 16), kept=(1, 1, 1), ddt=np.float64, dinit=3.0,
             ins=[(np.float64, None, False)], exact=False),
        dict(name="sum_init_conj_c64", f="ident", op="+", initop="conj", dims=(12, 10, 16), kept=(1, 1, 1), ddt=np.complex64, dinit=3.0 + 1.0j,
             ins=[(np.complex64, None, False)], exact=False),
        # the other reduction operators
        dict(name="min_f64", f="ident", op="min", initop=None, dims=(33, 17, 24), kept=(1, 1, 1), ddt=np.float64, dinit=np.inf,
             ins=[(np.float64, None, True)], exact=True),
        dict(name="max_of_diff_f32", f="scaled_diff", op="max", initop=None, dims=(33, 17, 24), kept=(1, 1, 1), ddt=np.float32, dinit=-np.inf,
             ins=[(np.float32, None, False), (np.float32, None, False)], exact=True),
        dict(name="all_positive", f="gt0", op="&", initop=None, dims=(20, 20, 16), kept=(1, 1, 1), ddt=np.uint8, dinit=1,
             ins=[(np.float64, None, False)], exact=True, shift=5.0, plant=((3, 4, 15), -1.0)),   # one counterexample, in the LAST rank's slab
        dict(name="any_positive", f="gt0", op="|", initop=None, dims=(20, 20, 16), kept=(1, 1, 1), ddt=np.uint8, dinit=0,
             ins=[(np.float64, None, False)], exact=True, shift=-5.0, plant=((0, 0, 0), 1.0)),   # one witness, in the FIRST rank's slab
        dict(name="sum_int64", f="ident", op="+", initop=None, dims=(50, 20, 16), kept=(1, 1, 1), ddt=np.int64, dinit=11,
             ins=[(np.int64, None, False)], exact=True),
        dict(name="max_int16", f="ident", op="max", initop=None, dims=(50, 20, 16), kept=(1, 1, 1), ddt=np.int16, dinit=-32768,
             ins=[(np.int16, None, True)], exact=True),           # 16-bit destination: 32-bit staging through the collective
        dict(name="sum_int16_wraps", f="ident", op="+", initop=None, dims=(50, 20, 16), kept=(1, 1, 1), ddt=np.int16, dinit=5,
             ins=[(np.int16, None, False)], exact=True),
        # partial reductions
        dict(name="partial_kept_split", f="ident", op="+", initop=None, dims=(40, 24, 32), kept=(1, 24, 32), ddt=np.float64, dinit=1.0,
             ins=[(np.float64, None, False)], exact=False),       # a kept dim is long enough: disjoint destination slabs, no collective
        dict(name="partial_reduced_split_strided_dest", f="abs2", op="+", initop=("scale", 2.0), dims=(small, small - 1, 64), kept=(small, small - 1, 1),
             ddt=np.float64, dinit=1.5, ins=[(np.float64, None, True)], exact=False, dest_strides=True),
        dict(name="partial_reduced_split_max", f="ident", op="max", initop=None, dims=(small, 48, small - 1), kept=(small, 1, small - 1),
             ddt=np.float32, dinit=-np.inf, ins=[(np.float32, (0, 1, 2), False)], exact=True, dest_strides=True),
        # a map: every rank computes its slab of the destination, nothing is exchanged
        dict(name="map_permutedims", f="ident", op=None, initop=None, dims=(24, 32, 40), kept=(24, 32, 40), ddt=np.float64, dinit=0.0,
             ins=[(np.float64, (2, 0, 1), False)], exact=True),
    ]
    if world == 2:  # with two ranks every kept extent >= 2 can be split: only complete reductions take the collective
        out = [c for c in out if not c["name"].startswith("partial_reduced_split")]
    return out


def gen_input(case, k, seed):
    """full (unsharded) input k of a case, column-major, in the shape the f-call sees AFTER its permutation is applied"""
    dt, perm, _ = case["ins"][k]
    dims = case["dims"]
    rng = np.random.default_rng(seed * 100 + k)
    shape = dims if perm is None else tuple(dims[perm.index(i)] for i in range(len(dims)))  # parent shape: view = permutedims(parent, perm)
    if np.issubdtype(dt, np.integer):
        info = np.iinfo(dt)
        a = rng.integers(info.min // 4, info.max // 4, size=shape, dtype=dt)
    elif np.issubdtype(dt, np.complexfloating):
        a = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
    else:
        a = rng.standard_normal(shape).astype(dt) if case["dims"][0] != 4096 else (rng.random(shape, dtype=np.float32) * 2 - 1)
    if "shift" in case:
        a = a + dt(case["shift"])
    if "plant" in case and k == 0:
        a[case["plant"][0]] = case["plant"][1]
    return np.asfortranarray(a)


def dest_layout(case):
    """(parent length, strides over the full box, offset) of the destination: dense over the kept elements, or strided with gaps"""
    kept, dims = case["kept"], case["dims"]
    st, s = [], (2 if case.get("dest_strides") else 1)
    for n, d in zip(kept, dims):
        st.append(0 if (n == 1 and d != 1) else s)
        if not (n == 1 and d != 1):
            s *= n
            if case.get("dest_strides"):
                s += 1
    return s + 3, tuple(st), (3 if case.get("dest_strides") else 0)


def expected(case, inputs):
    """NumPy truth over the full problem (float64 / exact integer accumulation), shaped like the kept box"""
    views = [a if p is None else np.transpose(a, p) for a, (_, p, _) in zip(inputs, case["ins"])]
    fname, op = case["f"], case["op"]
    acc_t = np.complex128 if np.issubdtype(case["ddt"], np.complexfloating) else (np.float64 if np.issubdtype(case["ddt"], np.floating) else None)
    cast = (lambda v: v.astype(acc_t)) if acc_t is not None else (lambda v: v.astype(np.int64) if np.issubdtype(v.dtype, np.integer) else v)
    v = [cast(x) for x in views]
    val = {"ident": lambda: v[0], "abs2": lambda: np.abs(v[0]) ** 2 if acc_t is np.complex128 else v[0] * v[0], "mul": lambda: v[0] * v[1],
           "gt0": lambda: (views[0] > 0), "scaled_diff": lambda: 2 * v[0] - v[1]}[fname]()
    if op is None:
        return np.asarray(val).astype(case["ddt"])
    axes = tuple(i for i, (n, d) in enumerate(zip(case["kept"], case["dims"])) if n == 1 and d != 1)
    d0 = case["dinit"]
    init = case["initop"]
    if init is None or init == "identity":
        start = d0
    elif init == "zero":
        start = 0
    elif init == "conj":
        start = np.conj(d0)
    elif init[0] == "scale":
        start = d0 * init[1]
    else:
        start = init[1]
    red = {"+": np.sum, "min": np.min, "max": np.max, "&": np.all, "|": np.any}[op]
    r = red(val, axis=axes, keepdims=True)
    if op == "+":
        r = r + start
    elif op == "min":
        r = np.minimum(r, start)
    elif op == "max":
        r = np.maximum(r, start)
    elif op == "&":
        r = np.logical_and(r, start != 0)
    else:
        r = np.logical_or(r, start != 0)
    if np.issubdtype(case["ddt"], np.integer) and r.dtype != np.bool_:
        bits = np.dtype(case["ddt"]).itemsize * 8
        r = (r.astype(np.int64) & ((1 << bits) - 1) if bits < 64 else r.astype(np.int64))
        if bits < 64:
            r = np.where(r >= (1 << (bits - 1)), r - (1 << bits), r)
    return np.asarray(r).astype(case["ddt"])
