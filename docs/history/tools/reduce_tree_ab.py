#!/usr/bin/env python3
"""Split partial reductions: second launch (reduce_tree = 0) against the two-level in-launch fold (reduce_tree = 1024)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import strided_jl_amd as S


def mk(shape, dt):
    t = torch.randn(int(torch.tensor(shape).prod()), dtype=dt, device="cuda")
    st, s = [], 1
    for d in shape:
        st.append(s); s *= d
    return S.StridedView(t, shape, tuple(st), 0)


def time_plan(plan, reps=200):
    cur = lambda: int(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.execute(cur())
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            plan.execute(cur())
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


if __name__ == "__main__":
  for dims, dt in (((100, 90, 80, 7), torch.float32), ((512, 384, 64), torch.float32), ((33, 100000), torch.float64), ((7, 3, 250000), torch.float32)):
      A = mk(dims, dt)
      for k in range(1, len(dims)):
          for rd in itertools.combinations(range(len(dims)), k):
              out = A.similar(size=tuple(1 if d in rd else n for d, n in enumerate(dims)))
              arrs = S.promoteshape(dims, out, A)
              row = []
              desc = ""
              for tree in (0, 1024):
                  S.set_option("reduce_tree", tree)
                  p = S.make_plan(lambda x: x, "+", None, dims, arrs)
                  desc = p.describe()
                  row.append(time_plan(p))
              S.set_option("reduce_tree", 0)
              if "split=1 " in desc + " ":
                  continue
              b = p.algorithmic_bytes
              print("sum %-18s dims=%-10s two launches %7.2f us | in-launch tree %7.2f us | %5.2f -> %5.2f TB/s | %s" %
                    (dims, rd, row[0], row[1], b / row[0] * 1e-6, b / row[1] * 1e-6, desc[desc.find("form="):desc.find("algbytes")]))
