#!/bin/bash
mkdir -p gpurun_out/c23
timeout 900 python -m pytest tests/test_gpu_fuzz_families.py -q -x -m gpu -k "reduce" > gpurun_out/c23/fuzz.log 2>&1
tail -4 gpurun_out/c23/fuzz.log
timeout 800 python tools/reduce_sweep.py > gpurun_out/c23/reduce_sweep.txt 2>&1
grep -B1 -A3 "reduce_seg=" gpurun_out/c23/reduce_sweep.txt | grep -v "^--" | grep "^float\|^complex\|reduce_seg\|defaults" | head -80
