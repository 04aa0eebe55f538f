#!/bin/bash
# round 5, call 24: the randomised comparison on a library-owned stream, harness synchronised as S.Stream's contract asks
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
{
OWN_STREAM=1 timeout 600 python tools/fuzz_more.py 61000 500 2>&1 | tail -5
BIG=1 OWN_STREAM=1 timeout 600 python tools/fuzz_more.py 63000 120 2>&1 | tail -5
} | tee gpurun_out/call24_fuzz.txt
