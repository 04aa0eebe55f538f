"""Runtime compilation of f-programs (csrc/smr_jit.cpp), CPU-only part: the generated functor
text, and that the embedded kernel sources compile with hiprtc for gfx950 for every kernel
family (no device needed; the GPU parity tests then run the same code objects)."""
import numpy as np
import pytest

import strided_jl_amd as S

fn = S.fn


def _v(shape, dtype=np.float64):
    return S.StridedView(np.zeros(shape, dtype=dtype, order="F"))


def test_generated_functor_mirrors_the_f_program():
    A, B, C = _v((64, 64)), _v((64, 64)), _v((64, 64))
    plan = S.make_plan(lambda a, c: a * 2 + c / 3 - 1, None, None, A.size, (B, A, C))
    src = plan.jit_source()
    assert "static constexpr int NIN = 2;" in src
    assert "typedef double JT;" in src
    # postfix order: a[0] 2 * a[1] 3 / + 1 -   constants are kernel arguments (k.c[2i], k.c[2i+1] = re, im of
    # constant i): the text -- the key of the compiled-code cache -- depends on the program's structure only
    body = [ln.strip() for ln in src.splitlines() if ln.strip().startswith("const JT v")]
    assert body[0].endswith("= a[0];") and "k.c[0]" in body[1] and "bin(34, v0, v1)" in body[2]
    assert body[3].endswith("= a[1];") and "k.c[2]" in body[4] and "bin(35, v3, v4)" in body[5]
    assert "bin(32, v2, v5)" in body[6] and "k.c[4]" in body[7] and "bin(33, v6, v7)" in body[8]
    other = S.make_plan(lambda a, c: a * 7.5 + c / 0.1 - 3, None, None, A.size, (B, A, C)).jit_source()
    assert other == src  # a different scalar does not make a different program
    # select(cond, then, else) pops three
    plan = S.make_plan(lambda a, c: fn.select(a < c, a, c), None, None, A.size, (B, A, C))
    assert "truthy(v2) ? v3 : v4" in plan.jit_source()


def _orbit_views(shape, dtype, perms):
    a = _v(shape, dtype)  # ONE buffer, permuted views: the ORBIT family
    return (a.similar(),) + tuple(a.permutedims(q) for q in perms)


CASES = {
    "stream_f64": lambda: (lambda a, c: a * 2 + c / 3 - 1, None, (_v((256, 256)), _v((256, 256)), _v((256, 256)))),
    "stream_c32": lambda: (lambda a, c: fn.conj(a) * c - 1j, None,
                           (_v((256, 256), np.complex64), _v((256, 256), np.complex64), _v((256, 256), np.complex64))),
    "stream_mixed": lambda: (lambda a, c: a * c - 0.5, None, (_v((256, 256)), _v((256, 256), np.float32), _v((256, 256), np.float32))),
    "tiled_f32": lambda: (lambda a, c: a - c, None, (_v((256, 256), np.float32), _v((256, 256), np.float32).permutedims((1, 0)), _v((256, 256), np.float32))),
    "tiled_big_f64": lambda: (lambda a, b, c, d: a * b - c * d, None,
                              (_v((32,) * 4),) + tuple(_v((32,) * 4).permutedims(q) for q in [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)])),
    "tiled_ragged_c64": lambda: (lambda a: fn.exp(a) * 2, None, (_v((100, 70), np.complex128), _v((70, 100), np.complex128).permutedims((1, 0)))),
    "orbit_f64": lambda: (lambda a, b, c, d: a * b - c * d, None, _orbit_views((32,) * 4, np.float64, [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)])),
    "orbit_c32": lambda: (lambda a, b: fn.conj(a) * b - np.complex64(1j), None, _orbit_views((256, 256), np.complex64, [(0, 1), (1, 0)])),
    "generic_f64": lambda: (lambda a: fn.sqrt(fn.abs(a)), None, (_v((7, 9, 5)), _v((5, 9, 7)).permutedims((2, 1, 0)))),
    "reduce_all_f32": lambda: (lambda a: fn.sin(a) * a, "+", None),
    "reduce_part_f64": lambda: (fn.sin, "+", None),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_every_family_compiles_for_gfx950(name):
    f, op, arrays = CASES[name]()
    if name == "reduce_all_f32":
        x = _v((64, 64, 16), np.float32)
        o = x.similar(size=(1,))
        arrays = S.promoteshape(x.size, o.sreshape((1, 1, 1)), x)
    elif name == "reduce_part_f64":
        x = _v((32, 16, 32, 8))
        o = x.similar(size=(32, 1, 32, 1))
        arrays = S.promoteshape(x.size, o, x)
    if name.startswith("generic"):
        S.set_option("force_family", 1)
    try:
        plan = S.make_plan(f, op, None, arrays[1].size if op else arrays[0].size, arrays)
    finally:
        S.set_option("force_family", 0)
    d = plan.describe()
    assert "f=prog" in d or "(mixed)" in d, d
    fam = name.split("_")[0]
    assert f"family={fam}" in d, d
    before = S.get_option("jit_failures")
    n = plan.jit_compile()
    assert n > 1000, f"{name}: no code object ({d})"
    assert S.get_option("jit_failures") == before


def test_native_functors_and_the_interpreter_need_no_compilation():
    A, B = _v((128, 128)), _v((128, 128))
    assert S.make_plan(lambda a: a, None, None, A.size, (B, A.permutedims((1, 0)))).jit_compile() == 0
    assert S.make_plan(lambda a, c: a + c, None, None, A.size, (B, A, A.permutedims((1, 0)))).jit_compile() == 0
    S.set_option("jit", 0)
    try:
        assert S.make_plan(lambda a: a * a - a, None, None, A.size, (B, A)).jit_compile() == 0
    finally:
        S.set_option("jit", 1)


def test_disk_cache_is_opt_in_and_reused(tmp_path, monkeypatch):
    """$SMR_JIT_CACHE_DIR: code objects are published there and a later compilation of the same
    source (same library build) is served from disk instead of running the compiler."""
    monkeypatch.setenv("SMR_JIT_CACHE_DIR", str(tmp_path))
    A, B = _v((96, 96)), _v((96, 96))
    f = lambda a: (a * a - a * 0.123456789) * a + a  # noqa: E731  (a program no other test compiles: constants do not distinguish programs)
    c0 = S.get_option("jit_compiles")
    n1 = S.make_plan(f, None, None, A.size, (B, A)).jit_compile()
    assert n1 > 0 and S.get_option("jit_compiles") == c0 + 1
    files = list(tmp_path.glob("smr_*.co"))
    assert len(files) == 1 and files[0].stat().st_size == n1


def test_missing_compiler_helper_is_reported_not_fatal(monkeypatch):
    """Without the helper the dry compile reports SMR_EUNSUPPORTED (on a GPU the launchers fall back to
    the interpreter after a one-time warning); nothing crashes and native functors are unaffected."""
    monkeypatch.setenv("SMR_JITC", "/nonexistent/smr_jitc")
    A, B = _v((80, 80)), _v((80, 80))
    f0 = S.get_option("jit_failures")
    plan = S.make_plan(lambda a: fn.tanh(a) * a - fn.cos(a) * 0.987654321, None, None, A.size, (B, A))   # a program no other test compiles
    with pytest.raises(S.UnsupportedOnDevice):
        plan.jit_compile()
    assert S.get_option("jit_failures") == f0 + 1
    assert S.make_plan(lambda a: a, None, None, A.size, (B, A.permutedims((1, 0)))).jit_compile() == 0
