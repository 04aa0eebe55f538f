#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py -q -x -k "rows_that_are_not" 2>&1 | tail -8
timeout 300 python tools/stream_ua_ab.py > $O/stream_ua_ab.txt 2>&1; cat $O/stream_ua_ab.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -n 4 2>&1 | tail -4
