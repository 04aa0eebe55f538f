#!/bin/bash
mkdir -p gpurun_out/c29
timeout 600 python tools/flat2_ab.py > gpurun_out/c29/flat2_ab.txt 2>&1
cat gpurun_out/c29/flat2_ab.txt | tail -52
timeout 900 python -m pytest tests/test_gpu_fuzz_families.py -q -x -m gpu -k "flat" > gpurun_out/c29/fuzz.log 2>&1
tail -4 gpurun_out/c29/fuzz.log
