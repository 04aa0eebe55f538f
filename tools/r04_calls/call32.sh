#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/flat_ab.py > $O/flat_ab.txt 2>/dev/null; cut -c1-150 $O/flat_ab.txt | grep -v amdgpu
timeout 600 python tools/flat2_ab.py > $O/flat2_ab.txt 2>/dev/null; cut -c1-200 $O/flat2_ab.txt | grep -v amdgpu
