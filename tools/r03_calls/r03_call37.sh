#!/bin/bash
mkdir -p gpurun_out/c37
timeout 50 python tools/flat2_ab.py --batched > gpurun_out/c37/flat2_batched.txt 2>&1
tail -14 gpurun_out/c37/flat2_batched.txt
