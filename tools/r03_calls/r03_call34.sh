#!/bin/bash
mkdir -p gpurun_out/c34
timeout 600 python tools/flat_nary_ab.py > gpurun_out/c34/flat_nary_ab.txt 2>&1
cat gpurun_out/c34/flat_nary_ab.txt | tail -32
timeout 900 python -m pytest tests/test_gpu_fuzz_families.py -q -x -m gpu -k "flat" > gpurun_out/c34/fuzz.log 2>&1
tail -4 gpurun_out/c34/fuzz.log
