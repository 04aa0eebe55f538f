#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/perf_sanity.py > $O/perf_sanity3.txt 2>/dev/null; wc -l $O/perf_sanity3.txt
timeout 600 python -m pytest tests/test_gpu_flatb.py tests/test_gpu_round4.py -q -n 2 2>&1 | tail -2
