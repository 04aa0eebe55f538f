#!/bin/bash
mkdir -p gpurun_out/c35
timeout 170 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py tests/test_integer_class.py tests/test_jit.py tests/test_julia_shim.py tests/test_mixed_precision.py -m gpu -q -x > gpurun_out/c35/tests.log 2>&1
tail -4 gpurun_out/c35/tests.log
