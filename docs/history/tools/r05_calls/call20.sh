#!/bin/bash
# round 5, call 20: narrow-integer re-wrapping on the device (compiled functors + interpreter) and the integer class suite
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_integer_typing.py tests/test_integer_class.py tests/test_jit.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/call20_int.txt
