#!/bin/bash
# r04 call 22: two-sided FLAT with evenly cut long leads: parity, then A/B (every row verified against torch)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py -q -x -k ragged 2>&1 | tail -5
timeout 900 python tools/flat2_long_ab.py > $O/flat2_long_ab.txt 2>&1; grep -v amdgpu.ids $O/flat2_long_ab.txt | cut -c1-200
