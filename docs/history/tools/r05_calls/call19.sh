#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/tiled_ua_ab.py > $O/tiled_ua_ab.txt 2>&1; cat $O/tiled_ua_ab.txt | cut -c1-330
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round4.py tests/test_gpu_flatb.py tests/test_gpu_round5.py tests/test_gpu_fuzz_families.py -q -x -n 4 2>&1 | tail -6
