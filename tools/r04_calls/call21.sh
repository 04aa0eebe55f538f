#!/bin/bash
# r04 call 21: two-sided FLAT with evenly cut long leads against TILED
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python tools/flat2_long_ab.py > $O/flat2_long_ab.txt 2>&1; grep -v amdgpu.ids $O/flat2_long_ab.txt | cut -c1-200
