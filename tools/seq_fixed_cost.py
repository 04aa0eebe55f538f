#!/usr/bin/env python3
"""Where do the fixed microseconds of a short replay go?  (The driver times bench.py with --steps 20.)
Prints, for the bench step (32^4 f64): wall clock of torch.cuda.synchronize() on an idle device, of smr_seq_run(K) + smr_seq_wait for
K = 1..2000 (and the library's own doorbell -> completion figure), and the same region bracketed as bench.py does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import strided_jl_amd as S


def main():
    n = 32
    dev = torch.device("cuda", 0)
    tA = torch.randn(n ** 4, dtype=torch.float64, device=dev)
    tB, tC = torch.empty_like(tA), torch.empty_like(tA)
    st = (1, n, n * n, n ** 3)
    A, B, C = (S.StridedView(t, (n,) * 4, st, 0) for t in (tA, tB, tC))
    perms = [(0, 1, 2, 3), (1, 2, 3, 0), (2, 3, 0, 1), (3, 0, 1, 2)]
    p2 = S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0))))
    p3 = S.make_plan(lambda a, b, c, d: a + b + c + d, None, None, A.size, (C,) + tuple(A.permutedims(p) for p in perms))
    cur = lambda: int(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    pc = time.perf_counter

    def best(fn, reps=30):
        b = 1e30
        for _ in range(reps):
            t = pc(); fn(); b = min(b, pc() - t)
        return b * 1e6

    torch.cuda.synchronize()
    print("perf_counter pair                      : %7.2f us" % best(lambda: None))
    print("torch.cuda.synchronize(), idle device  : %7.2f us" % best(torch.cuda.synchronize))
    st = S.Stream()  # library-owned: smr_seq_run returns after the doorbells, no holding kernel on a HIP stream
    variants = [("default (3 queues: perm | sum/2 | sum/2, acquire by need)", {}),
                ("... first acquire at agent scope (experiment)", {"first_acquire": 1}),
                ("... last release at agent scope (experiment)", {"last_release": 1}),
                ("... both (experiment)", {"first_acquire": 1, "last_release": 1}),
                ("2 queues, r04 fences", {"queues": 2, "slices": 1, "acquire": 1}),
                ("1 queue", {"queues": 1})]
    for name, opts in variants:
        q = S.Sequence().add(p2).add(p3)
        for k, v in opts.items():
            q.set(k, v)
        q.run(5, st.handle); q.wait()
        print("%s | %s" % (name, q.info()))
        for K in (1, 2, 5, 10, 20, 50, 100, 500, 2000):
            def region():
                q.run(K, st.handle); q.wait()
            w = best(region, 15)
            lib = float(q.info().split("last_replay_us=")[1])

            def bracketed():
                q.run(K, st.handle); q.wait(); torch.cuda.synchronize()
            wb = best(bracketed, 15)

            def run_only():
                q.run(K, st.handle)
            tr = 1e30
            for _ in range(10):
                t = pc(); q.run(K, st.handle); tr = min(tr, pc() - t); q.wait()
            print("  K=%5d  run+wait %9.2f us (%7.3f /step) | run() alone returns after %6.2f us | library doorbell->done %9.2f us | + torch.cuda.synchronize %9.2f us (%7.3f /step)"
                  % (K, w, w / K, tr * 1e6, lib, wb, wb / K))
        del q
    # the graph form of r3 for comparison
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        p2.execute(cur()); p3.execute(cur())
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g, stream=side):
        for _ in range(20):
            p2.execute(cur()); p3.execute(cur())
    g.replay(); torch.cuda.synchronize()

    def greg():
        g.replay(); torch.cuda.synchronize()
    w = best(greg, 15)
    print("hipGraph of 20 steps: replay + torch.cuda.synchronize %9.2f us (%7.3f /step)" % (w, w / 20))
    print("torch.cuda.synchronize(), idle, after graph streams exist: %7.2f us" % best(torch.cuda.synchronize))


main()
