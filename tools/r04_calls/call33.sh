#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu -rf -p no:cacheprovider -v > $O/pytest_gpu_full.txt 2>&1
grep -n "Fatal\|Segmentation\|Current thread" -A12 $O/pytest_gpu_full.txt | head -60 | cut -c1-200
grep -n "PASSED\|FAILED" $O/pytest_gpu_full.txt | tail -3 | cut -c1-200
tail -3 $O/pytest_gpu_full.txt | cut -c1-300
