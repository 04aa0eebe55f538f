#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python tools/flat2_long_ab.py > $O/flat2_long_ab2.txt 2>&1; grep -v amdgpu.ids $O/flat2_long_ab2.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_round4.py -q -k ragged 2>&1 | tail -2
