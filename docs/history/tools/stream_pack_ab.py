#!/usr/bin/env python3
"""STREAM family: one row segment per workgroup (stream_pack_rows = 0) against the packed form for rows of 129 .. 128*U vectors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import strided_jl_amd as S
from reduce_tree_ab import mk, time_plan  # noqa: E402

for dt in (torch.float64, torch.float32):
    for n0 in (130, 200, 257, 300, 384, 400, 513, 561, 700, 1000, 1025, 1500, 2047):
        for outer in ((65, 129), (33, 31)):
            shape = (n0,) + outer
            A = mk(shape, dt)
            B = mk((n0, outer[1], outer[0]), dt)
            row = []
            for pack in (0, 1):
                S.set_option("stream_pack_rows", pack)
                p = S.make_plan(lambda x: x, None, None, B.size, (B, A.permutedims((0, 2, 1))))
                row.append(time_plan(p))
            S.set_option("stream_pack_rows", 1)
            b = p.algorithmic_bytes
            print("%-8s permutedims %-18s (0,2,1): one row/wg %6.2f us | packed %6.2f us | %5.2f -> %5.2f TB/s | %5.1f MiB" %
                  (str(dt).replace("torch.", ""), shape, row[0], row[1], b / row[0] * 1e-6, b / row[1] * 1e-6, b / 2 ** 20))
