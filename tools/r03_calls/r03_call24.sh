#!/bin/bash
mkdir -p gpurun_out/c24
timeout 800 python tools/reduce_sweep.py > gpurun_out/c24/reduce_sweep.txt 2>&1
grep -c . gpurun_out/c24/reduce_sweep.txt
