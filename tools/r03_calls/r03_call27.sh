#!/bin/bash
mkdir -p gpurun_out/c27
timeout 900 python tools/perf_sanity.py > gpurun_out/c27/perf_sanity.txt 2>&1
head -45 gpurun_out/c27/perf_sanity.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "stale" > gpurun_out/c27/tests.log 2>&1
tail -3 gpurun_out/c27/tests.log
