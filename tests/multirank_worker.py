"""One rank of tests/test_gpu_multirank.py: `python multirank_worker.py RANK WORLD UIDHEX OUTDIR [cpu]`.
All ranks share GPU 0; the collective is tests/libfake_rccl.so (SMR_RCCL_LIB), everything else is the product path:
smr_comm_init -> smr_mapreduce_sharded_ex (smr_shard_ex, fill_neutral on ranks != 0, the HIP kernels, gather -> ncclAllReduce ->
scatter).  Every rank holds only what its role needs: block-partitioned inputs (`local`) exist as the rank's slab alone."""
import ctypes as C
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, uid, outdir = int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3]), sys.argv[4]
    import torch
    import strided_jl_amd as S
    from strided_jl_amd import _lib as L
    from strided_jl_amd import distributed as D
    import multirank_cases as MC

    torch.cuda.set_device(0)
    D.comm_init(world, rank, uid)
    # with four ranks every call goes through a library-owned stream (eager direct dispatch): the gather and the scatter around the
    # collective are then submitted by the library itself, and the collective is foreign work it has to fence for
    lib_stream = S.Stream() if world == 4 else None
    got_rank, got_world = D.comm_rank()
    res = {"comm_rank": np.array([got_rank, got_world]), "library": np.array(D.comm_library())}

    def colmajor(shape):
        st, s = [], 1
        for d in shape:
            st.append(s)
            s *= d
        return tuple(st)

    for ci, case in enumerate(MC.cases(world)):
        dims = case["dims"]
        full = [MC.gen_input(case, k, ci) for k in range(len(case["ins"]))]
        # the logical problem over host views, only to ask smr_shard_ex which slab this rank owns
        plen, dstr, doff = MC.dest_layout(case)
        hdest = S.StridedView(np.zeros(plen, dtype=case["ddt"]), dims, dstr, doff)
        hins = []
        for a, (_, perm, _) in zip(full, case["ins"]):
            v = S.StridedView(a)
            hins.append(v if perm is None else v.permutedims(perm))
        hp, keep = S.build_problem(MC.F[case["f"]], case["op"], case["initop"], dims, (hdest,) + tuple(hins), stream=0)
        sub = L.smr_problem()
        need, sdim, start, stop = C.c_int(0), C.c_int(-1), C.c_int64(0), C.c_int64(0)
        mask = sum(1 << (k + 1) for k, (_, _, loc) in enumerate(case["ins"]) if loc)
        L.check(L.load().smr_shard_ex(C.byref(hp), world, rank, mask, C.byref(sub), C.byref(need), C.byref(sdim), C.byref(start), C.byref(stop)))
        sdim, start, stop = sdim.value, start.value, stop.value
        # device operands
        dparent = torch.full((plen,), 77, dtype=getattr(torch, np.dtype(case["ddt"]).name), device="cuda")
        dest = S.StridedView(dparent, dims, dstr, doff)
        kept_view = S.StridedView(dparent, case["kept"], tuple(s if s else 1 for s in dstr), doff)
        S.copyto_(kept_view, case["dinit"])  # every rank starts from the same destination content
        ins, local = [], [False]
        for a, (_, perm, loc) in zip(full, case["ins"]):
            if loc:
                assert perm is None
                sl = [slice(None)] * a.ndim
                sl[sdim] = slice(start, stop)
                slab = np.asfortranarray(a[tuple(sl)])
                t = torch.from_numpy(slab.ravel(order="F").copy()).cuda()
                ins.append(S.StridedView(t, dims, colmajor(slab.shape), 0))  # logical box over the slab's memory
            else:
                t = torch.from_numpy(a.ravel(order="F").copy()).cuda()
                v = S.StridedView(t, a.shape, colmajor(a.shape), 0)
                ins.append(v if perm is None else v.permutedims(perm))
            local.append(bool(loc))
        del full
        torch.cuda.synchronize()  # torch prepared the operands on its own stream
        before = (S.get_option("launches"), S.get_option("allreduces"), S.get_option("allreduces_inplace"))
        D.comm_mapreduce_sharded_(MC.F[case["f"]], case["op"], case["initop"], dims, (dest,) + tuple(ins), local=tuple(local),
                                  stream=lib_stream.handle if lib_stream else None)
        if lib_stream:
            lib_stream.synchronize()
        torch.cuda.synchronize()
        res["dest_%d" % ci] = dparent.cpu().numpy()
        res["meta_%d" % ci] = np.array([need.value, sdim, start, stop])
        # kernel launches / all-reduces / in-place all-reduces this call issued
        res["counts_%d" % ci] = np.array([S.get_option("launches") - before[0], S.get_option("allreduces") - before[1], S.get_option("allreduces_inplace") - before[2]])
        # Float32 / ComplexF32 sums once more with the ranks' partials crossing as Float64 (option allreduce_f64)
        if case["op"] == "+" and case["ddt"] in (np.float32, np.complex64) and need.value:
            S.copyto_(kept_view, case["dinit"])
            torch.cuda.synchronize()
            S.set_option("allreduce_f64", 1)
            D.comm_mapreduce_sharded_(MC.F[case["f"]], case["op"], case["initop"], dims, (dest,) + tuple(ins), local=tuple(local),
                                      stream=lib_stream.handle if lib_stream else None)
            S.set_option("allreduce_f64", 0)
            if lib_stream:
                lib_stream.synchronize()
            torch.cuda.synchronize()
            res["dest64_%d" % ci] = dparent.cpu().numpy()
    if lib_stream:
        res["eager_launches"] = np.array(S.get_option("eager_launches"))
        lib_stream.close()
    D.comm_destroy()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)


if __name__ == "__main__":
    try:
        main()
    except Exception:  # noqa: BLE001
        traceback.print_exc()
        sys.exit(1)
