#!/bin/bash
# round 5, call 25: reversed-destination initop on the device; more fuzz seeds on both kinds of stream
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
{
timeout 300 python -m pytest tests/test_oracle_numpy.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
BIG=1 timeout 900 python tools/fuzz_more.py 64000 250 2>&1 | tail -4
BIG=1 OWN_STREAM=1 timeout 900 python tools/fuzz_more.py 65000 250 2>&1 | tail -4
} | grep -v amdgpu.ids | tee gpurun_out/call25_fuzz.txt
