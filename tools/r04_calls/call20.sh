#!/bin/bash
# r04 call 20: batched FLAT form
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flatb.py -q -x 2>&1 | tail -12
timeout 600 python tools/flat2_ab.py --batched > $O/flat2_batched.txt 2>&1; grep -v amdgpu.ids $O/flat2_batched.txt | cut -c1-250
