#!/bin/bash
mkdir -p gpurun_out/c20
timeout 800 python tools/reduce_sweep.py --full > gpurun_out/c20/reduce_sweep.txt 2>&1
tail -5 gpurun_out/c20/reduce_sweep.txt
