#!/usr/bin/env python3
"""bench.py -- headline benchmark of the strided map/reduce hot path on MI355X.

Metric (BASELINE.json): GB/s effective HBM (and % of roofline) for @strided permutedims! +
broadcast on 32^4 Float64.  One STEP = the two README workloads on one pair of 32^4 f64 arrays,
inputs resident in HBM:
    (C2)  permutedims!(B, A, (4,3,2,1))                         README.md:98
    (C3)  C .= A_p1 .+ A_p2 .+ A_p3 .+ A_p4  (4 permuted views) README.md:104   (into a THIRD array C, chosen so that the two
          operations of a step share nothing but the read-only input A and may overlap; the README's session writes both
          into B -- that write-after-write chain is timed as well and reported on the same line: "step_same_destination")
Algorithmic bytes per step = 2 launches x (8 MiB read + 8 MiB written) = 33,554,432 B
(every distinct array counted once).

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU.  Started under torch.distributed.run (WORLD_SIZE set) the process is a rank;
started plain with --gpus N > 1 it re-executes itself under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` and the N ranks rendezvous on 127.0.0.1.
Every rank runs the step on its own arrays (independent objects, no data-path collective -> "weak"
scaling) and the value is the whole-job aggregate.  The reduce path (C4, RCCL all-reduce of the per-shard partial) is
reported under "extra".

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event timed inside this run) and
"cpu_baseline" (the oracle = C++ restatement of the reference's algorithm, on this host's cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def colmajor_view(S, t, shape):
    st, s = [], 1
    for d in shape:
        st.append(s)
        s *= d
    return S.StridedView(t, shape, tuple(st), 0)


def event_time_ms(torch, fn, iters):
    """Average per-call time of fn() over `iters` back-to-back calls, HIP events on the current
    stream (the stream the kernels are launched on)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def graph_two_chains(torch, fn_a, fn_b, reps):
    """`reps` steps whose two INDEPENDENT launches (they only share a read-only input) run as two stream-ordered
    chains inside one hipGraph: the launches that write one output array stay ordered among themselves, the two
    chains are forked once and joined once.  fn_a / fn_b take the raw stream handle to launch on."""
    g = torch.cuda.CUDAGraph()
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        fn_a(int(main.cuda_stream))
        fn_b(int(main.cuda_stream))
    torch.cuda.current_stream().wait_stream(main)
    with torch.cuda.graph(g, stream=main):
        side.wait_stream(main)
        for _ in range(reps):
            fn_a(int(main.cuda_stream))
            fn_b(int(side.cuda_stream))
        main.wait_stream(side)
    return g


def graph_of(torch, fn, reps):
    """Capture `reps` calls of fn() into one hipGraph (removes host launch overhead; the
    kernels are ~microseconds, a Python-side launch is not)."""
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn()
    return g


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(S, budget_s=24.0, full=True):
    """The reference algorithm's CPU path (oracle: fuse/order/blocks/threaded bisection/kernel
    restated in C++) timed on this box's host cores: the headline 32^4 f64 step, and -- as bounded
    samples -- the other BASELINE.json configs and the README / benchmarks/benchtests.jl extras."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    fn = S.fn
    cores = os.cpu_count() or 1
    threads = sorted({1, min(4, cores), cores})
    t_end = time.perf_counter() + budget_s

    def best_of(problem, nt, max_reps, t_stop):
        best, reps = 1e30, 0
        while reps < 2 or (time.perf_counter() < t_stop and reps < max_reps):
            t0 = time.perf_counter()
            oraclelib.mapreduce(problem, nt)
            best = min(best, time.perf_counter() - t0)
            reps += 1
        return best, reps

    n = 32
    rng = np.random.default_rng(1234)
    A = S.StridedView(np.asfortranarray(rng.standard_normal((n,) * 4)))
    B = A.similar()
    C = A.similar()
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    p2, k2 = S.build_problem(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))), stream=0)
    p3, k3 = S.build_problem(lambda a, b, c, d: a + b + c + d, None, None, A.size,
                             (C,) + tuple(A.permutedims(p) for p in perms), stream=0)
    out = {}
    for nt in threads:
        t_stop = min(t_end, time.perf_counter() + budget_s / 8)
        b2, r2 = best_of(p2, nt, 200, t_stop)
        b3, r3 = best_of(p3, nt, 200, t_stop + budget_s / 16)
        out[nt] = dict(threads=nt, permutedims_ms=b2 * 1e3, broadcast4_ms=b3 * 1e3,
                       gbs=2 * 16777216 / (b2 + b3) / 1e9, reps=min(r2, r3))
    best = max(out.values(), key=lambda d: d["gbs"])

    # the other configs (bounded samples; algorithmic bytes = distinct operand footprints)
    configs = {}

    def sample(name, what, problem, algbytes, keep, only=None):
        # only=(): the big configs run at the largest thread count (a single-threaded pass over 4 GiB would eat the
        # whole budget) -- 2 repetitions, min taken
        rows = []
        for nt in (threads if only is None else [threads[-1]]):
            b, r = best_of(problem, nt, 20 if only is None else 2, min(t_end, time.perf_counter() + budget_s / 16))
            rows.append({"threads": nt, "ms": round(b * 1e3, 3), "GB/s": round(algbytes / b / 1e9, 2), "reps": r})
        configs[name] = {"sample": what, "algorithmic_bytes": algbytes, "best_GB/s": max(r["GB/s"] for r in rows), "by_threads": rows}
        del keep

    m = 4000  # configs[0]: the reference's own CPU-runnable case, full size
    A1 = S.StridedView(np.asfortranarray(rng.standard_normal((m, m))))
    B1 = A1.similar()
    p, k = S.build_problem(lambda x, y: (x + y) / 2, None, None, (m, m), (B1, A1, A1.adjoint()), stream=0)
    sample("c1_symmetrise_4000_f64", "B .= (A .+ A')./2, 4000x4000 Float64, full size (README.md:60-90)", p, 2 * 8 * m * m, k)
    del A1, B1
    m = 1000
    A1 = S.StridedView(np.asfortranarray(rng.standard_normal((m, m))))
    B1 = A1.similar()
    p, k = S.build_problem(lambda x: 3 * x, None, None, (m, m), (B1, A1.adjoint()), stream=0)
    sample("readme_3A'_1000_f64", "B .= 3 .* A', 1000x1000 Float64 (README.md:79-83)", p, 2 * 8 * m * m, k)
    for q, name in (((1, 2, 3, 0), "perm_2341_32^4_f64"), ((2, 3, 0, 1), "perm_3412_32^4_f64")):
        p, k = S.build_problem(lambda x: x, None, None, A.size, (B, A.permutedims(q)), stream=0)
        sample(name, "permutedims!(B, A, %s), 32^4 Float64 (benchmarks/benchtests.jl:40-42)" % (tuple(i + 1 for i in q),), p, 2 * 8 * n ** 4, k)
    # configs[3] and configs[4] at FULL size; the 4 GiB input is generated slab by slab
    nz = 64 if full else 2
    xa = np.empty((4096, 4096, nz), dtype=np.float32, order="F")
    for z in range(nz):
        xa[:, :, z] = rng.random((4096, 4096), dtype=np.float32) * 2 - 1
    X = S.StridedView(xa)
    o = S.StridedView(np.zeros(1, dtype=np.float32), X.size, (0, 0, 0), 0)
    p, k = S.build_problem(fn.abs2, "+", None, X.size, (o, X), stream=0)
    sample("c4_mapreduce_abs2_f32", "mapreduce(abs2,+) on 4096x4096x%d Float32 (%s configs[3])" % (nz, "full" if full else "1/32 of"),
           p, 4 * 4096 * 4096 * nz + 4, k, only=() if full else None)
    del X, xa
    ny = 8192 if full else 1024
    Y = S.StridedView(np.asfortranarray(rng.random((8192, ny), dtype=np.float32)))
    Z = Y.similar()
    p, k = S.build_problem(lambda a: a * fn.exp(-2 * a) + fn.sin(a * a), None, None, Y.size, (Z, Y), stream=0)
    sample("c5_expr_f32", "B .= A.*exp.(-2A) .+ sin.(A.*A) on 8192x%d Float32 (%s configs[4])" % (ny, "full" if full else "1/8 of"),
           p, 2 * 4 * 8192 * ny, k, only=() if full else None)
    return {
        "value": round(best["gbs"], 3), "unit": "GB/s", "cores": best["threads"], "kind": "port",
        "cpu_model": _cpu_model(), "host_cores": cores,
        "sample": "min over %d repetitions of the same 32^4 f64 step (permutedims! + 4-way broadcast), "
                  "oracle = C++ restatement of the reference algorithm; host has %d cores (%s)" % (best["reps"], cores, _cpu_model()),
        "by_threads": [out[k] for k in sorted(out)],
        "configs": configs,
        "readme_4threads_gbs": {"permutedims": 14.07, "broadcast4": 6.00, "hardware": "unstated (README.md:143-153)"},
    }


# The shape of the JSON line, in one place: --dry (the CPU test of the N-rank path) prints it, the real run asserts it.
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_replay", "closing_device_sync_us", "higher_is_better", "scaling", "degraded",
             "vs_baseline", "dtype", "data", "rehearsal", "config", "frac_of_hbm_peak", "step_same_destination", "roofline")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "frac_rocprof_avg", "us_rocprof_avg", "rocprof_file",
                 "per_kernel", "library_replay_alone", "note", "hbm_cold")
EXTRA_KEYS = {"c1": ("c1_symmetrise_4000_f64",), "hbm128": ("permutedims_128^4_f64", "broadcast4_128^4_f64"),
              "cold": ("permutedims_32^4_f64_cold", "permutedims_32^4_f64_cold_seq_2_queues", "broadcast4_32^4_f64_cold", "broadcast4_32^4_f64_cold_seq_2_queues"),
              "c5": ("c5_expr_8192_f32",), "c4": ("c4_mapreduce_abs2_4096x4096x64_f32",),
              "sharded": ("c1_symmetrise_sharded", "broadcast4_sharded", "broadcast4_128^4_sharded")}


def extra_keys(world, extras):
    """Names the secondary workloads put under "extra" for a job of `world` ranks (the sharded map/permute legs run for N > 1, or on request)."""
    only = None if extras == "all" else set(extras.split(","))
    names = []
    for leg, keys in EXTRA_KEYS.items():
        if leg == "sharded" and world == 1 and (only is None or "sharded" not in only):
            continue
        if only is None or leg in only:
            names.extend(keys)
    return names


def relaunch_as_ranks(ngpus):
    """`python bench.py --gpus N` started without a launcher: become the launcher (one rank per GPU)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """--dry: no GPU.  The ranks rendezvous over gloo, agree on the world size and rank 0 prints the line's
    launch-related fields (CPU test of the `--gpus N` path)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([1.0, float(rank)], dtype=torch.float64)
        dist.all_reduce(t)
        seen, ranksum = int(t[0].item()), int(t[1].item())
    else:
        seen, ranksum = 1, 0
    assert seen == world and ranksum == world * (world - 1) // 2
    if rank == 0:
        print(json.dumps({"dry": True, "n_gpus": seen, "requested_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "backend": "gloo" if world > 1 else "none",
                          # the shape the real line will have for this job (asserted by the real run before it prints)
                          "line_keys": list(LINE_KEYS) + ([] if args.no_cpu or seen > 1 else ["cpu_baseline"]) + ([] if args.no_extra else ["extra"]),
                          "roofline_keys": list(ROOFLINE_KEYS), "extra_keys": [] if args.no_extra else extra_keys(seen, args.extras),
                          "timed_region": "barrier | per-rank clock: K steps + local device sync | barrier, then MAX over ranks"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--extra-timeout", type=int, default=300, help="N > 1: seconds the secondary workloads may take before the headline is printed without them")
    ap.add_argument("--extras", default="all", help="comma-separated subset of the secondary workloads: c1,hbm128,cold,c5,c4,sharded (default: all; `sharded` = the map/permute configs cut into --gpus slabs, runs by default only for N > 1)")
    ap.add_argument("--step-mode", choices=("seq", "seq1", "inorder", "chains"), default="seq",
                    help="seq: the library's own replay of the recorded step (smr_seq: AQL packets on its HSA queues, one queue per "
                         "dependency component -- the step's two operations are independent, both only read A -- results of in-order "
                         "execution); seq1: the same replay on ONE queue, in order; inorder: hipGraph, the step's two launches on one "
                         "stream; chains: hipGraph, one stream-ordered chain per output array")
    ap.add_argument("--dry", action="store_true", help="rendezvous only (gloo, no GPU): checks that --gpus N yields N ranks")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(relaunch_as_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if args.dry:
        return dry_run(args, world, rank)

    import numpy as np
    import torch
    import strided_jl_amd as S

    # Rendezvous: torch.distributed over RCCL ("nccl").  With $SMR_RCCL_LIB set -- the library's collective is then a stand-in
    # (tests/libfake_rccl.so: N processes sharing ONE GPU) -- the ranks meet over gloo instead and all land on the devices there are:
    # the rehearsal of the whole N-rank bench path (launcher, communicator, sharded config 4, JSON assembly) on a one-GPU box.
    rehearsal = bool(os.environ.get("SMR_RCCL_LIB"))
    ndev = max(1, torch.cuda.device_count())
    local = local % ndev if rehearsal else local
    torch.cuda.set_device(local)          # before the process group: RCCL binds its communicator to the current device
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    red_dev = torch.device("cpu") if rehearsal else dev   # where the ranks' scalars are reduced (gloo: host tensors)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n = 32
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    tA = torch.randn(n ** 4, dtype=torch.float64, device=dev, generator=g)
    tB = torch.empty_like(tA)
    tC = torch.empty_like(tA)
    A, B, C = (colmajor_view(S, t, (n,) * 4) for t in (tA, tB, tC))
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    plan2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    plan3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size,
                        (C,) + tuple(A.permutedims(p) for p in perms))
    bytes2, bytes3 = plan2.algorithmic_bytes, plan3.algorithmic_bytes
    assert bytes2 == bytes3 == 2 * 8 * n ** 4

    def cur():
        return int(torch.cuda.current_stream().cuda_stream)

    def step():
        s = cur()
        plan2.execute(s)
        plan3.execute(s)

    # correctness guard inside the bench: outputs must equal the torch permutes exactly
    a4 = tA.reshape((n,) * 4)  # row-major view of the same memory: index order reversed
    # column-major (i1,i2,i3,i4) <-> torch index [i4,i3,i2,i1]
    ref2 = a4.permute(3, 2, 1, 0).contiguous().reshape(-1)
    cm = lambda p: a4.permute(*[3 - p[3 - i] for i in range(4)])  # noqa: E731
    ref3 = (((cm(perms[0]) + cm(perms[1])) + cm(perms[2])) + cm(perms[3])).contiguous().reshape(-1)

    def check_outputs(when):
        torch.cuda.synchronize()
        assert torch.equal(tB, ref2), "permutedims! result is wrong (%s)" % when
        assert torch.equal(tC, ref3), "fused 4-way broadcast result is wrong (%s)" % when

    step()
    check_outputs("first step")

    K, W = args.steps, args.warmup
    for _ in range(W):
        step()
    use_graph = not args.no_graph and args.step_mode in ("inorder", "chains")
    use_seq = not args.no_graph and args.step_mode in ("seq", "seq1")
    seq_info = None
    if use_seq:
        # the recorded step, replayed by the library (csrc/smr_seq.cpp).  One smr_seq_run(K) = K steps; it returns when the
        # replay has completed on devices without stream-side waits, smr_seq_wait covers the others.
        # The replay is issued on a library-owned stream (smr_stream_create): smr_seq_run returns after the doorbells and needs no
        # holding kernel on a HIP stream; smr_seq_wait is the host-side wait; torch.cuda.synchronize() in barrier() covers the device.
        seq = S.Sequence().add(plan2).add(plan3)
        if args.step_mode == "seq1":
            seq.set("queues", 1)
        seq_stream = S.Stream()
        tB.zero_(); tC.zero_()
        torch.cuda.synchronize()
        seq.run(max(1, W), seq_stream.handle)  # untimed: builds the packets, creates the queues
        seq.wait()
        check_outputs("sequence warm-up")
    if use_graph:
        chunk = min(K, 500)
        while K % chunk:
            chunk -= 1
        if args.step_mode == "chains":
            gstep = graph_two_chains(torch, plan2.execute, plan3.execute, chunk)
        else:
            gstep = graph_of(torch, step, chunk)
        nrep = K // chunk
        gstep.replay()  # untimed: first replay uploads the graph
    if use_seq:  # untimed rehearsal of exactly the timed call (host code paths, queue doorbells and signals warm)
        seq.run(K, seq_stream.handle)
        seq.wait()
    tB.zero_(); tC.zero_()  # the timed region must (re)produce both outputs
    stream_handle = seq_stream.handle if use_seq else cur()
    # Timed region (N ranks): collective barrier + device sync, THEN every rank's own clock around its K steps and its own device
    # sync; the MAX over ranks is taken afterwards.  (Until round 5 the closing dist.barrier() sat inside the region: a collective
    # of ~2 ms around 0.1 ms of work -- profiles/r05_bench_gpus8_rehearsal.json.  The ranks share no data on this path, so the
    # slowest rank's local time IS the job's time.)
    barrier()
    t0 = time.perf_counter()
    if use_seq:
        seq.run(K, stream_handle)
        seq.wait()       # host-side wait on the replay's completion signals: the K steps are done when it returns (the replay does not
        #                  run on a HIP stream, a device sync cannot see it) ...
        t_done = time.perf_counter()
        torch.cuda.synchronize()   # ... and the device sync the contract brackets the region with finds nothing left to wait for
        t_sync = time.perf_counter() - t_done
    elif use_graph:
        for _ in range(nrep):
            gstep.replay()
        torch.cuda.synchronize()
    else:
        for _ in range(K):
            step()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    check_outputs("after the timed region")  # bit-exact, AFTER timing: the overlapped replay has in-order results
    if use_seq:
        seq_info = seq.info()
        # (absent when the sequence replays through HIP: under a profiler, or with a kernel that needs scratch)
        replay_us = float(seq_info.split("last_replay_us=")[1].split()[0]) if "last_replay_us=" in seq_info else None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = world * K * (bytes2 + bytes3) / dt / 1e9

    # per-kernel launch durations, HIP events on the launch stream (graph replay of one kernel
    # type back to back: includes the dependent-launch gap, excludes host overhead)
    reps = 500
    s = cur()
    g2 = graph_of(torch, lambda: plan2.execute(cur()), reps)
    g3 = graph_of(torch, lambda: plan3.execute(cur()), reps)
    g2.replay(); g3.replay()
    torch.cuda.synchronize()
    import statistics
    s2 = [event_time_ms(torch, g2.replay, 2) / reps for _ in range(9)]   # 9 samples of 2 x 500 launches
    s3 = [event_time_ms(torch, g3.replay, 2) / reps for _ in range(9)]
    ms2, ms3 = min(s2), min(s3)
    med2, med3 = statistics.median(s2), statistics.median(s3)
    # the step both ways, long graphs, HIP events (the headline value above uses --step-mode over the driver's K)
    gi = graph_of(torch, step, reps)
    gc = graph_two_chains(torch, plan2.execute, plan3.execute, reps)
    gi.replay(); gc.replay()
    torch.cuda.synchronize()
    step_us = {"inorder": round(min(event_time_ms(torch, gi.replay, 2) for _ in range(7)) / reps * 1e3, 3),
               "chains": round(min(event_time_ms(torch, gc.replay, 2) for _ in range(7)) / reps * 1e3, 3)}
    del gi, gc

    def seq_us(queues, nsteps):
        """wall clock (run + wait) per step of `nsteps` replays of the recorded step; best of 7"""
        q = S.Sequence().add(plan2).add(plan3)
        for k, v in queues.items():
            q.set(k, v)
        st = S.Stream()
        q.run(5, st.handle); q.wait()
        best = 1e30
        for _ in range(7):
            torch.cuda.synchronize()
            t = time.perf_counter()
            q.run(nsteps, st.handle); q.wait()
            best = min(best, time.perf_counter() - t)
        info = q.info()
        del q
        st.close()
        return round(best / nsteps * 1e6, 3), info
    step_us["seq_one_queue"], _ = seq_us({"queues": 1}, 2 * reps)
    step_us["seq_queue_per_component"], _ = seq_us({"queues": 2, "slices": 1}, 2 * reps)
    step_us["seq_queue_per_component_r04_fences"], _ = seq_us({"queues": 2, "slices": 1, "acquire": 1}, 2 * reps)
    step_us["seq_default"], sinfo = seq_us({}, 2 * reps)
    # the same step issued EAGERLY, call by call from Python, on a library-owned stream (eager direct dispatch: the library submits each
    # launch itself on the hardware queue its data dependencies select) against eager calls on a HIP stream
    def eager_us(handle, sync, nsteps=1000):
        for _ in range(20):
            plan2.execute(handle); plan3.execute(handle)
        sync()
        best = 1e30
        for _ in range(5):
            t = time.perf_counter()
            for _ in range(nsteps):
                plan2.execute(handle); plan3.execute(handle)
            sync()
            best = min(best, time.perf_counter() - t)
        return round(best / nsteps * 1e6, 3)
    torch.cuda.synchronize()
    lib_stream = S.Stream()
    step_us["eager_library_stream"] = eager_us(lib_stream.handle, lib_stream.synchronize)
    lib_stream.close()
    hs = torch.cuda.Stream()
    step_us["eager_hip_stream"] = eager_us(int(hs.cuda_stream), hs.synchronize)
    check_outputs("after the eager steps")
    step_us["note"] = ("inorder/chains: hipGraph replay, HIP events; seq_*: smr_seq replay of 1000 steps, host wall clock around "
                       "smr_seq_run + smr_seq_wait (the replay does not run on a HIP stream); eager_*: 1000 steps issued call by call from Python, wall clock "
                       "including the final synchronisation")
    check_outputs("after the long replays")
    # the same kernels replayed ALONE by the library (its own AQL packets: self-released launch, no fences inside the replay), host
    # wall clock / launches: what a launch costs where the library dispatches itself (a sequence, a library-owned stream)
    def replay_alone_us(plan, opts, nl=2000):
        q = S.Sequence().add(plan)
        for k, v in opts.items():
            q.set(k, v)
        stq = S.Stream()
        q.run(50, stq.handle); q.wait()
        best = 1e30
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            q.run(nl, stq.handle); q.wait()
            best = min(best, time.perf_counter() - t)
        del q
        stq.close()
        return round(best / nl * 1e6, 3)
    replay = {name: {"one_queue_us": replay_alone_us(pl, {"queues": 1, "slices": 1}), "cut_in_two_us": replay_alone_us(pl, {"queues": 2, "slices": 2})}
              for name, pl in (("permutedims", plan2), ("broadcast4", plan3))} if use_seq else {}
    # The README's own form of the step (/root/reference README.md:92-105): BOTH statements write B.  A write-after-write chain: the
    # second launch may not overtake the first, nothing overlaps.  Timed the same two ways as the headline's form.
    plan3b = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(p) for p in perms))

    def step_b():
        s_ = cur()
        plan2.execute(s_)
        plan3b.execute(s_)
    step_b()
    torch.cuda.synchronize()
    assert torch.equal(tB, ref3), "same-destination step: B must hold the 4-way sum"
    gb = graph_of(torch, step_b, reps)
    gb.replay()
    torch.cuda.synchronize()
    same_graph_us = min(event_time_ms(torch, gb.replay, 2) for _ in range(7)) / reps * 1e3
    del gb
    same_seq_us = None
    if use_seq:
        qb = S.Sequence().add(plan2).add(plan3b)
        stb = S.Stream()
        qb.run(5, stb.handle); qb.wait()
        same_seq_us = 1e30
        for _ in range(7):
            torch.cuda.synchronize()
            t = time.perf_counter()
            qb.run(2 * reps, stb.handle); qb.wait()
            same_seq_us = min(same_seq_us, (time.perf_counter() - t) / (2 * reps) * 1e6)
        torch.cuda.synchronize()
        assert torch.equal(tB, ref3), "same-destination replay: B must hold the 4-way sum"
        del qb
        stb.close()
    same_us = min(x for x in (same_graph_us, same_seq_us) if x is not None)
    step_same_destination = {
        "us_per_step": round(same_us, 3), "GB/s": round((bytes2 + bytes3) / same_us / 1e3, 1),
        "frac_of_hbm_peak": round((bytes2 + bytes3) / same_us / 1e3 / HBM_PEAK_GBS, 4),
        "hipgraph_in_order_us": round(same_graph_us, 3), "library_replay_us": round(same_seq_us, 3) if same_seq_us is not None else None,
        "note": "permutedims!(B, A, (4,3,2,1)) then B .= sum of 4 permuted views of A: both statements into B as in the reference's README "
                "session (README.md:92-105) -- a write-after-write chain, no overlap possible; `value` above is the step with the sum "
                "written to a third array C"}
    plan2.execute(cur())   # B holds the permutation again (the headline arrays are verified once more below)
    torch.cuda.synchronize()
    del plan3b

    dom = ("broadcast4", ms3, bytes3, plan3) if ms3 >= ms2 else ("permutedims", ms2, bytes2, plan2)
    # rocprofv3's average duration of the dominant kernel, from the tracked summary of the same command (profiles/): the profiler
    # serialises sub-5-us dispatches, so this is the pessimistic witness next to the HIP-event minimum above
    rocprof = {"file": None, "us_avg": None, "frac": None}
    try:
        import glob
        import re
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_kernel_trace_stats.txt")))
        if cands:
            want_k = ("k_orbit_", "FAdd4") if dom[0] == "broadcast4" else ("k_tiled_map", "FIdent")  # (k_orbit_pair since round 6, k_orbit_map before)
            for line in open(cands[-1]):
                f = line.split()
                if len(f) >= 5 and f[0].isdigit() and all(w in line for w in want_k):
                    rocprof = {"file": "profiles/" + os.path.basename(cands[-1]), "us_avg": round(float(f[1]) / 1e3, 3),
                               "frac": round(dom[2] / float(f[1]) / HBM_PEAK_GBS, 4)}
                    break
    except Exception:  # noqa: BLE001 -- a missing / unreadable summary must not cost the line
        pass
    achieved = dom[2] / (dom[1] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom[0])
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "us_per_launch": round(dom[1] * 1e3, 3),
        # the same kernel by rocprofv3's AVERAGE over the tracked kernel trace of this command (profiler-serialised dispatches)
        "frac_rocprof_avg": rocprof["frac"], "us_rocprof_avg": rocprof["us_avg"], "rocprof_file": rocprof["file"],
        "per_kernel": {
            "permutedims": {"us": round(ms2 * 1e3, 3), "us_median": round(med2 * 1e3, 3), "GB/s": round(bytes2 / (ms2 * 1e-3) / 1e9, 1),
                            "frac": round(bytes2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "plan": plan2.describe()},
            "broadcast4": {"us": round(ms3 * 1e3, 3), "us_median": round(med3 * 1e3, 3), "GB/s": round(bytes3 / (ms3 * 1e-3) / 1e9, 1),
                           "frac": round(bytes3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "plan": plan3.describe()},
        },
        "library_replay_alone": replay,
        "note": "32^4 f64 = 8 MiB in + 8 MiB out per launch: the working set is L2/Infinity-Cache resident across "
                "back-to-back launches, so 'achieved' is effective (algorithmic bytes / launch time), not DRAM traffic. 'achieved' / 'frac' / "
                "'per_kernel' time the kernel as ANY caller launches it through HIP (HIP events on the launch stream, hipGraph of 500 launches: "
                "the figure rocprofv3 can be compared with); 'library_replay_alone' is the same kernel replayed alone by the library's own "
                "dispatch (host wall clock; HIP events cannot see those queues)",
    }

    extra = {}
    stuck = False
    if not args.no_extra:
        def run_extra():
            try:
                torch.cuda.set_device(local)
                if os.environ.get("BENCH_SIMULATE_STUCK_RANK") == str(rank):   # test hook for the watchdog below
                    time.sleep(1e6)
                return secondary(S, torch, np, dev, world, rank, event_time_ms, graph_of, colmajor_view, cur,
                                 only=None if args.extras == "all" else set(args.extras.split(",")), red_dev=red_dev)
            except Exception as e:  # noqa: BLE001 -- the headline line must still be printed
                return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if world == 1:
            extra = run_extra()
        else:
            # N ranks: the secondary workloads include the sharded config 4, whose all-reduce the library issues on its own RCCL
            # communicator -- a collective that does not complete (a rank that failed, a bootstrap that cannot connect) must not take
            # the headline line with it: the legs run beside a watchdog, and a rank that is still stuck when it fires reports that and
            # leaves without the final barrier
            import threading
            box = {}
            th = threading.Thread(target=lambda: box.update(extra=run_extra()), daemon=True)
            th.start()
            th.join(timeout=args.extra_timeout)
            stuck = th.is_alive()
            extra = ({"error": "the secondary workloads did not finish within %d s on rank %d (a collective that did not complete?); the headline "
                               "above was measured before them" % (args.extra_timeout, rank)} if stuck else box.get("extra", {}))

    # HBM-cold fractions of the two headline kernels (rotating through more arrays than the Infinity Cache holds): the bandwidth truth
    # next to the cache-resident figures, promoted out of `extra`
    if isinstance(extra, dict):
        cold = {k: {"us": v.get("us"), "frac": v.get("frac_of_8TBs")} for k, v in extra.items() if k.endswith("_cold") or "_cold_" in k}
        roofline["hbm_cold"] = cold or None
    if rank == 0:
        out = {
            "metric": "GB/s effective HBM for @strided permutedims!+broadcast, 32^4 fp64",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 6),
            # the library's own clock around the same K steps: first doorbell -> completion signals observed (no Python, no torch sync)
            "ms_per_step_replay": round(replay_us / K * 1e-3, 6) if (use_seq and replay_us is not None) else None,
            # of ms_per_step * steps: the closing torch.cuda.synchronize() alone, after smr_seq_wait had already seen the completion signals
            "closing_device_sync_us": round(t_sync * 1e6, 1) if use_seq else None,
            "higher_is_better": True, "scaling": "weak",
            # a secondary workload failed or a rank's watchdog fired: the headline was measured before them, but this is not a clean run
            "degraded": bool(stuck or (isinstance(extra, dict) and "error" in extra)),
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rehearsal": ("$SMR_RCCL_LIB is set: %d ranks over gloo sharing %d GPU(s), the collective is a stand-in -- a rehearsal of the N-rank "
                          "path, not a measurement" % (world, ndev)) if rehearsal else None,
            "config": {"workload": "configs[1]+configs[2]: permutedims!(B,A,(4,3,2,1)) then C .= sum of 4 permuted views of A, 32x32x32x32 Float64, one "
                                   "set of arrays (A, B, C) per GPU.  C is a THIRD array chosen so that the two statements share only the read-only A "
                                   "and overlap; the README writes both into B -- that form is `step_same_destination` on this line",
                       "algorithmic_bytes_per_step": bytes2 + bytes3,
                       "launch": ("smr_seq replay (AQL packets on the library's HSA queues), " +
                                  ("one queue per dependency component, the heavier chain (the 4-way sum) cut into two block ranges: the step's two "
                                   "operations only share the read-only input A and overlap; acquire fences only where the sequence reads what it "
                                   "writes (nowhere here); results of in-order execution, verified bit-exactly after the timed region" if args.step_mode == "seq"
                                   else "one queue, in recorded order") + " | " + str(seq_info)) if use_seq else
                                 ("hipGraph, " + ("two stream-ordered chains (one per output array; both operations only read A), one fork / one join per graph"
                                                  if args.step_mode == "chains" else "in order on one stream")) if use_graph else "eager",
                       "step_us_long_graph": step_us,
                       "parallelism": "replicas x%d (independent arrays per rank)" % world},
            "frac_of_hbm_peak": round(value / world / HBM_PEAK_GBS, 4),
            "step_same_destination": step_same_destination,
            "roofline": roofline,
        }
        if not args.no_cpu and world == 1:  # timed on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(S)
        if extra:
            out["extra"] = extra
        # the line has the shape --dry announces (tests/test_distributed_cpu.py checks that one on the CPU)
        assert set(LINE_KEYS) <= set(out), sorted(set(LINE_KEYS) - set(out))
        assert set(ROOFLINE_KEYS) <= set(roofline), sorted(set(ROOFLINE_KEYS) - set(roofline))
        if extra and "error" not in extra:
            missing = [k for k in extra_keys(world, args.extras) if k not in extra]
            assert not missing, "secondary workloads missing from the line: %s" % missing
        print(json.dumps(out), flush=True)
    if world > 1:
        if not stuck:   # (a peer may be stuck: the closing barrier gets a watchdog of its own)
            import threading
            th = threading.Thread(target=lambda: (dist.barrier(), dist.destroy_process_group()), daemon=True)
            th.start()
            th.join(timeout=60)
            stuck = th.is_alive()
        if stuck:
            # a rescued run is not a clean one: the line carries extra.error, and the exit status says so too
            sys.stdout.flush()
            os._exit(3)


def secondary(S, torch, np, dev, world, rank, event_time_ms, graph_of, colmajor_view, cur, only=None, red_dev=None):
    """The other BASELINE.json configs, short runs (not the headline value)."""
    res = {}
    fn = S.fn
    red_dev = red_dev or dev

    def want(name):
        return only is None or name in only

    def timed(plan, reps=50):
        g = graph_of(torch, lambda: plan.execute(cur()), reps)
        g.replay()
        torch.cuda.synchronize()
        ms = min(event_time_ms(torch, g.replay, 2) for _ in range(3)) / reps
        return ms

    def rec(name, plan, ms):
        b = plan.algorithmic_bytes
        res[name] = {"us": round(ms * 1e3, 2), "GB/s": round(b / (ms * 1e-3) / 1e9, 1),
                     "frac_of_8TBs": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "plan": plan.describe()}

    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]

    def sec_c1():
        # C1 symmetrise 4000^2 f64
        m = 4000
        tA = torch.randn(m * m, dtype=torch.float64, device=dev)
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        p = S.make_plan(lambda x, y: (x + y) / 2, None, None, (m, m), (B, A, A.adjoint()))
        rec("c1_symmetrise_4000_f64", p, timed(p))

    def sec_hbm128():
        # true-HBM variant of the headline: 128^4 f64 (2 GiB in, 2 GiB out)
        n = 128
        tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
        p = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
        rec("permutedims_128^4_f64", p, timed(p, 5))
        perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
        p = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (B,) + tuple(A.permutedims(q) for q in perms))
        rec("broadcast4_128^4_f64", p, timed(p, 5))
        del tA, tB

    def sec_cold():
        # "cold" variant of the headline kernels: rotate through 40 distinct (A, B) pairs =
        # 640 MiB > the 256 MiB Infinity Cache, so every launch really reads HBM
        n, npair = 32, 40
        poolA = torch.randn(npair, n ** 4, dtype=torch.float64, device=dev)
        poolB = torch.empty_like(poolA)
        A, B = colmajor_view(S, poolA[0], (n,) * 4), colmajor_view(S, poolB[0], (n,) * 4)
        esz = poolA.element_size() * n ** 4
        for name, f, srcs in (("permutedims_32^4_f64_cold", lambda x: x, (A.permutedims((3, 2, 1, 0)),)),
                              ("broadcast4_32^4_f64_cold", lambda a, b, c, d: a + b + c + d, tuple(A.permutedims(q) for q in perms))):
            p = S.make_plan(f, None, None, A.size, (B,) + srcs)
            p.execute(cur())
            state = {"i": 0}

            def rot():
                i = state["i"] % npair
                state["i"] += 1
                p.execute(cur(), bases=[poolB.data_ptr() + i * esz] + [poolA.data_ptr() + i * esz] * len(srcs))

            g = graph_of(torch, rot, 4 * npair)
            g.replay()
            torch.cuda.synchronize()
            rec(name, p, min(event_time_ms(torch, g.replay, 2) for _ in range(3)) / (4 * npair))
            # the same 40 INDEPENDENT launches as a recorded sequence on two hardware queues (the boundary of one launch overlaps the next
            # pair's kernel; two queues measured best, profiles/r04_seq_vs_eager.txt): wall clock around smr_seq_run + smr_seq_wait
            q = S.Sequence()
            for i in range(npair):
                q.add(p, bases=[poolB.data_ptr() + i * esz] + [poolA.data_ptr() + i * esz] * len(srcs))
            q.set("queues", 2)
            q.set("async", 0)
            q.run(2, cur()); q.wait()
            best = 1e30
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                q.run(8, cur()); q.wait()
                best = min(best, time.perf_counter() - t0)
            rec(name + "_seq_2_queues", p, best / (8 * npair) * 1e3)
            del q
        del poolA, poolB

    def sec_c5():
        # C5 compute-bound map 8192^2 f32
        m = 8192
        tA = torch.rand(m * m, dtype=torch.float32, device=dev)
        tB = torch.empty_like(tA)
        A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
        p = S.make_plan(lambda a: a * fn.exp(-2 * a) + fn.sin(a * a), None, None, (m, m), (B, A))
        rec("c5_expr_8192_f32", p, timed(p, 20))

    def sec_c4():
        # C4 mapreduce(abs2,+) 4096x4096x64 f32, block-partitioned over the ranks on dim 3 (every rank holds ONLY its
        # slab), executed through the library's own multi-GPU entry point: smr_shard_ex -> local kernel ->
        # ONE ncclAllReduce (RCCL over xGMI, csrc/smr_comm.cpp), in place on the one-element destination
        from strided_jl_amd import distributed as D
        slab = 64 // world if 64 % world == 0 else 64
        tA = (torch.rand(4096 * 4096 * slab, dtype=torch.float32, device=dev) * 2 - 1)
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        dims = (4096, 4096, 64 if world > 1 and slab * world == 64 else slab)
        A = S.StridedView(tA, dims, (1, 4096, 4096 * 4096), 0)  # logical box over the slab's memory
        O = S.StridedView(out, dims, (0, 0, 0), 0)
        sharded = world > 1 and slab * world == 64
        if sharded:
            import torch.distributed as dist
            uid = [D.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            D.comm_init(world, rank, uid[0])
            rccl_rank, rccl_ranks = D.comm_rank()
            assert (rccl_rank, rccl_ranks) == (rank, world), "RCCL communicator disagrees with the launcher"

            def c4():
                out.zero_()
                D.comm_mapreduce_sharded_(fn.abs2, "+", None, dims, (O, A), local=(False, True))
            desc = "smr_mapreduce_sharded_ex (C ABI): shards=%d, slab-local input, ncclAllReduce(1 x f32) in place on the destination" % world
        else:
            p = S.make_plan(fn.abs2, "+", None, dims, (O, A))

            def c4():
                out.zero_()
                p.execute(cur())
            desc = p.describe()
        for _ in range(3):
            c4()
        torch.cuda.synchronize()
        got = float(out.item())
        # what ONE step issues: kernel launches of the library (the local reduction, on ranks != 0 the neutral fill) and all-reduces
        # -- the all-reduce runs in place on the one-element destination: no gather / scatter launches (csrc/smr_comm.cpp)
        c0 = (S.get_option("launches"), S.get_option("allreduces"), S.get_option("allreduces_inplace"))
        c4()
        torch.cuda.synchronize()
        per_step = {"library_kernel_launches": S.get_option("launches") - c0[0], "allreduces": S.get_option("allreduces") - c0[1],
                    "allreduces_in_place": S.get_option("allreduces_inplace") - c0[2]}
        ms = min(event_time_ms(torch, c4, 10) for _ in range(3))
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms = float(tt.item())
        b = 4 * 4096 * 4096 * slab * (world if sharded else 1)
        truth = float((tA.double() ** 2).sum().item())
        if sharded:
            tt = torch.tensor([truth], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt)
            truth = float(tt.item())
            D.comm_destroy()
        res["c4_mapreduce_abs2_4096x4096x64_f32"] = {
            "us": round(ms * 1e3, 2), "GB/s_total": round(b / (ms * 1e-3) / 1e9, 1), "shards": world if sharded else 1,
            "frac_of_8TBs_per_gpu": round(b / (world if sharded else 1) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "rel_err_vs_f64": abs(got - truth) / truth, "plan": desc,
            "collective": "RCCL ncclAllReduce(1 x f32) issued by libstrided_hip (smr_comm.cpp)" if sharded else "none",
            "rccl_ranks": rccl_ranks if sharded else 1, "per_step_rank0": per_step}

    def sec_sharded():
        # north_star: "large arrays are block-partitioned across the GPUs ...; map/permute stays embarrassingly parallel per shard".
        # The 4000^2 symmetrisation and the 4-way permuted sum (32^4, 128^4), cut along the destination's slowest dim into `world`
        # slabs (smr_shard_ex: the reference's own offset arithmetic, /root/reference/src/mapreduce.jl:203-222): every rank holds a
        # full replica of the source (a permuted source slab is a strided sub-box of the whole array) and writes only its slab of the
        # destination -- no collective on the data path.  Time = the slowest rank's; bytes = the whole problem's.
        import torch.distributed as dist
        from strided_jl_amd import distributed as D

        def run(name, f, dims, mk_arrays, reps):
            arrays = mk_arrays()
            sdims, sarrays, need, _ = D.shard(f, None, None, dims, arrays, world, rank)
            assert not need, "a map needs no all-reduce"
            p = S.make_plan(f, None, None, sdims, sarrays)
            ms = timed(p, reps)
            full = S.make_plan(f, None, None, dims, arrays)
            b = full.algorithmic_bytes
            if world > 1:
                tt = torch.tensor([ms], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ms = float(tt.item())
            res[name] = {"us": round(ms * 1e3, 2), "shards": world, "GB/s_total": round(b / (ms * 1e-3) / 1e9, 1),
                         "GB/s_per_gpu": round(b / world / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs_per_gpu": round(b / world / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "shard_dims": list(sdims), "collective": "none (replicated source, disjoint destination slabs)", "plan_of_this_rank": p.describe()}

        m = 4000
        g = torch.Generator(device=dev)
        g.manual_seed(4321)   # the same replica on every rank

        def sym_arrays():
            tA = torch.randn(m * m, dtype=torch.float64, device=dev, generator=g)
            tB = torch.empty_like(tA)
            A, B = colmajor_view(S, tA, (m, m)), colmajor_view(S, tB, (m, m))
            return (B, A, A.adjoint())
        run("c1_symmetrise_sharded", lambda x, y: (x + y) / 2, (m, m), sym_arrays, 50)
        for n, reps in ((32, 200), (128, 5)):
            def sum_arrays(n=n):
                tA = torch.randn(n ** 4, dtype=torch.float64, device=dev, generator=g)
                tB = torch.empty_like(tA)
                A, B = colmajor_view(S, tA, (n,) * 4), colmajor_view(S, tB, (n,) * 4)
                return (B,) + tuple(A.permutedims(q) for q in perms)
            run("broadcast4_sharded" if n == 32 else "broadcast4_128^4_sharded", lambda a, b, c, d: a + b + c + d, (n,) * 4, sum_arrays, reps)

    for name, sec in (("c1", sec_c1), ("hbm128", sec_hbm128), ("cold", sec_cold), ("c5", sec_c5), ("c4", sec_c4), ("sharded", sec_sharded)):
        if name == "sharded" and world == 1 and (only is None or "sharded" not in only):
            continue   # one GPU: the sharded legs are the unsharded ones (ask for them with --extras sharded)
        if want(name):
            sec()
    return res


if __name__ == "__main__":
    main()
