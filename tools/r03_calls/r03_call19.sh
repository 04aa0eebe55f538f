#!/bin/bash
# round 3, GPU call 19: one-launch split reductions -- parity + A/B timing
mkdir -p gpurun_out/c19
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "stale or flat_family or every_kernel_family or reduc or mapreduce or matmul or sum" > gpurun_out/c19/tests.log 2>&1
tail -5 gpurun_out/c19/tests.log
timeout 600 python -m pytest tests/test_gpu_fuzz_families.py tests/test_integer_class.py -q -x -m gpu > gpurun_out/c19/fuzz.log 2>&1
tail -3 gpurun_out/c19/fuzz.log
timeout 600 python tools/reduce_single_ab.py > gpurun_out/c19/reduce_single_ab.txt 2>&1
tail -70 gpurun_out/c19/reduce_single_ab.txt
