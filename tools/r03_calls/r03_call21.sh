#!/bin/bash
mkdir -p gpurun_out/c21
timeout 800 python tools/reduce_sweep.py > gpurun_out/c21/reduce_sweep.txt 2>&1
grep -c . gpurun_out/c21/reduce_sweep.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "stale or reduc or mapreduce or matmul or sum or golden or family" > gpurun_out/c21/tests.log 2>&1
tail -3 gpurun_out/c21/tests.log
timeout 600 python -m pytest tests/test_gpu_fuzz_families.py -q -x -m gpu -k "reduce" > gpurun_out/c21/fuzz.log 2>&1
tail -3 gpurun_out/c21/fuzz.log
