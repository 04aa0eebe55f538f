#!/usr/bin/env python3
"""Transposes of 1-, 2-, 4-byte element types (bit copies): bytes moved per workgroup matter (GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import strided_jl_amd as S  # noqa: E402
from bench import colmajor_view, event_time_ms  # noqa: E402


def cur():
    return int(torch.cuda.current_stream().cuda_stream)


def main():
    for dt in (torch.uint8, torch.int16, torch.int32, torch.float32, torch.float64):
        m = 16384
        t = torch.randint(0, 100, (m * m,), dtype=torch.int32, device="cuda").to(dt)
        o = torch.empty_like(t)
        A, B = colmajor_view(S, t, (m, m)), colmajor_view(S, o, (m, m))
        for tl in (0, 12):
            S.set_option("tile_log2", tl)
            plan = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims((1, 0))))
            plan.execute(cur())
            torch.cuda.synchronize()
            ms = min(event_time_ms(torch, lambda: plan.execute(cur()), 3) for _ in range(3))
            d = plan.describe()
            print(f"{str(dt)[6:]:8s} 16384^2 transpose tl={tl:2d} {ms * 1e3:9.1f} us {2 * t.element_size() * m * m / ms / 1e6:8.1f} GB/s | {d[d.find('tile='):d.find(' algbytes')]}")
        S.set_option("tile_log2", 0)
        tt = t.view(m, m)
        ms = min(event_time_ms(torch, lambda: tt.t().contiguous(), 3) for _ in range(3))
        print(f"{str(dt)[6:]:8s} torch .t().contiguous()      {ms * 1e3:9.1f} us {2 * t.element_size() * m * m / ms / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
