#!/bin/bash
# round 5, call 21: SMR_BOOL + narrow-integer re-wrapping on the device; Bool reductions; golden vectors and the parity list (dtype plumbing)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_integer_typing.py tests/test_integer_class.py tests/test_bool_reductions.py tests/test_jit.py tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/call21_int.txt
