"""GPU: csrc/smr_comm.cpp with MORE THAN ONE RANK on the hardware at hand.  2, 4 and 8 PROCESSES
share GPU 0; each initialises the library's communicator and calls smr_mapreduce_sharded_ex holding only its own slab of the
block-partitioned inputs.  The collective itself is tests/libfake_rccl.so -- a shared-memory stand-in selected through
$SMR_RCCL_LIB that exports the entry points libstrided_hip dlopens -- so everything AROUND ncclAllReduce is the product path:
smr_shard_ex's slab / offset rule for `local_ops`, fill_neutral on ranks != 0, initop applied once (rank 0), the gather of a strided
destination into staging, the scatter back, 16-bit destinations through 32-bit staging.  Reference analogue: the per-task partial
slots + fold of /root/reference/src/mapreduce.jl:153-170 and the task bisection of :195-227.  Expected values: NumPy over the full
problem (and the CPU oracle for the floating-point sums)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import multirank_cases as MC
import strided_jl_amd as S
from util import rtol, run_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "libfake_rccl.so")


def build_fake():
    if os.path.exists(FAKE) and os.path.getmtime(FAKE) >= os.path.getmtime(os.path.join(ROOT, "tests", "fake_rccl.cpp")):
        return
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "fake_rccl.cpp"), "-o", FAKE, "-lrt"],
                   check=True, capture_output=True)


def kept_elements(case, parent):
    plen, dstr, doff = MC.dest_layout(case)
    idx = np.full(case["kept"], doff, dtype=np.int64)
    for ax, (n, s) in enumerate(zip(case["kept"], dstr)):
        shape = [1] * len(case["kept"])
        shape[ax] = n
        idx = idx + (np.arange(n) * s).reshape(shape)
    return parent[idx], idx


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_sharing_one_gpu(world, tmp_path):
    build_fake()
    env = dict(os.environ, SMR_RCCL_LIB=FAKE)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rank 0's smr_comm_unique_id, shipped to the other ranks out of band (here: argv)
    uid = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from strided_jl_amd import distributed as D; "
                          "print(D.comm_unique_id().hex())" % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert uid.returncode == 0, uid.stderr[-2000:]
    uidhex = uid.stdout.strip().splitlines()[-1]
    worker = os.path.join(ROOT, "tests", "multirank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), uidhex, str(tmp_path)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    res = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]
    for r in range(world):
        assert tuple(res[r]["comm_rank"]) == (r, world)            # what the COMMUNICATOR says (ncclCommUserRank / ncclCommCount)
        assert str(res[r]["library"]).endswith("libfake_rccl.so")
    if world == 4:  # these ranks ran everything on a library-owned stream
        assert all(int(res[r]["eager_launches"]) > 20 for r in range(world))
    collectives = 0
    for ci, case in enumerate(MC.cases(world)):
        full = [MC.gen_input(case, k, ci) for k in range(len(case["ins"]))]
        want = MC.expected(case, full)
        tol = 0 if case["exact"] else 8 * rtol(np.float32 if np.dtype(case["ddt"]).itemsize <= 4 or case["ddt"] == np.complex64 else np.float64)
        need = int(res[0]["meta_%d" % ci][0])
        assert all(int(res[r]["meta_%d" % ci][0]) == need for r in range(world))
        if case["name"].startswith(("partial_kept", "map_")):
            assert need == 0, case["name"]
        elif case["op"] is not None:
            assert need == 1, case["name"]
        collectives += need
        # round 5: a dense kept destination in a type RCCL knows is all-reduced IN PLACE -- the call is the local kernel(s) (+ the
        # neutral fill on ranks != 0) + ONE all-reduce; strided destinations and 16-bit integers go through staging (two more launches)
        for r in range(world):
            launches, ar, ar_in = (int(v) for v in res[r]["counts_%d" % ci])
            assert ar == need, (case["name"], r, ar)
            if need:
                dense = not case.get("dest_strides")
                staged = np.dtype(case["ddt"]).itemsize == 2 or not dense
                assert ar_in == (0 if staged else 1), (case["name"], r, ar_in)
                if case["name"] == "c4_abs2_sum_f32":   # config 4: two reduction launches on rank 0 (+ the fill elsewhere), ONE collective, nothing else
                    assert launches <= (2 if r == 0 else 3), (case["name"], r, launches)
        for r in range(world):
            parent = res[r]["dest_%d" % ci]
            got, idx = kept_elements(case, parent)
            if need:  # every rank holds the complete result
                sel = (slice(None),) * got.ndim
            else:     # every rank computed its slab of the destination; the rest kept the initial content
                _, sdim, start, stop = (int(v) for v in res[r]["meta_%d" % ci])
                sel = tuple(slice(start, stop) if ax == sdim else slice(None) for ax in range(got.ndim))
                rest = np.ones(got.shape, dtype=bool)
                rest[sel] = False
                assert np.all(got[rest] == case["ddt"](case["dinit"])), (case["name"], r, "elements outside the rank's slab were touched")
            g, w = got[sel], want[sel]
            if tol == 0:
                assert np.array_equal(g, w), (case["name"], world, r)
            else:
                assert np.allclose(g, w, rtol=tol, atol=tol * float(np.max(np.abs(w)))), (case["name"], world, r, g.ravel()[:4], w.ravel()[:4])
            mask = np.ones(parent.shape, dtype=bool)
            mask[idx.ravel()] = False
            assert np.all(parent[mask] == 77), (case["name"], r, "memory between the destination's elements was written")
        if need and "dest64_%d" % ci in res[0].files:   # the Float64 crossing: same truth, bit-identical on every rank
            for r in range(world):
                g64, _ = kept_elements(case, res[r]["dest64_%d" % ci])
                assert np.allclose(g64, want, rtol=tol, atol=tol * float(np.max(np.abs(want)))), (case["name"], "allreduce_f64", r)
                assert np.array_equal(g64, kept_elements(case, res[0]["dest64_%d" % ci])[0])
        if need and not case["exact"]:
            for r in range(1, world):  # the collective leaves bit-identical results on every rank
                assert np.array_equal(kept_elements(case, res[r]["dest_%d" % ci])[0], kept_elements(case, res[0]["dest_%d" % ci])[0]), case["name"]
        # the CPU oracle on the unsharded problem, for the floating-point sums
        if case["op"] == "+" and not case["exact"] and case["dims"][0] != 4096:
            plen, dstr, doff = MC.dest_layout(case)
            hpar = np.full(plen, 77, dtype=case["ddt"])
            hdest = S.StridedView(hpar, case["dims"], dstr, doff)
            kv = S.StridedView(hpar, case["kept"], tuple(s if s else 1 for s in dstr), doff)
            kv.toarray()  # (view construction check)
            got0, idx = kept_elements(case, hpar)
            hpar[idx.ravel()] = case["dinit"]
            hins = []
            for a, (_, perm, _) in zip(full, case["ins"]):
                v = S.StridedView(a)
                hins.append(v if perm is None else v.permutedims(perm))
            run_oracle(MC.F[case["f"]], case["op"], case["initop"], case["dims"], (hdest,) + tuple(hins), 1)
            ora = kept_elements(case, hpar)[0]
            if need:
                dev = kept_elements(case, res[world - 1]["dest_%d" % ci])[0]
                assert np.allclose(dev, ora, rtol=tol, atol=tol * float(np.max(np.abs(ora)))), (case["name"], "vs oracle")
    assert collectives >= (14 if world > 2 else 12)
