#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/flat2_pair_ab.py > $O/flat2_pair_ab.txt 2>&1; grep -v amdgpu $O/flat2_pair_ab.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_fuzz_families.py -q -k "ragged or nary or flat" -n 2 2>&1 | tail -3
