#!/bin/bash
# round 5, call 23: more seeds of the randomised HIP-vs-oracle comparison, on HIP's stream and on a library-owned one
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
{
SMR_FUZZ_SEED_OFFSET=70000 timeout 900 python -m pytest tests/test_gpu_fuzz_families.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_more.py 60000 500 2>&1 | tail -5
OWN_STREAM=1 timeout 600 python tools/fuzz_more.py 61000 500 2>&1 | tail -5
BIG=1 timeout 600 python tools/fuzz_more.py 62000 120 2>&1 | tail -5
BIG=1 OWN_STREAM=1 timeout 600 python tools/fuzz_more.py 63000 120 2>&1 | tail -5
} | tee gpurun_out/call23_fuzz.txt
