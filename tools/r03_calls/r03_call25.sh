#!/bin/bash
mkdir -p gpurun_out/c25
timeout 800 python tools/reduce_sweep.py > gpurun_out/c25/reduce_sweep.txt 2>&1
grep -c . gpurun_out/c25/reduce_sweep.txt
timeout 900 python -m pytest tests/test_gpu_fuzz_families.py -q -x -m gpu -k "reduce" > gpurun_out/c25/fuzz.log 2>&1
tail -4 gpurun_out/c25/fuzz.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_integer_class.py -q -x -m gpu -k "stale or reduc or mapreduce or matmul or sum or golden or family or integer or int" > gpurun_out/c25/tests.log 2>&1
tail -3 gpurun_out/c25/tests.log
