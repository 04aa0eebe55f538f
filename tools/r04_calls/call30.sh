#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_round4.py -q -k "nary or ragged" 2>&1 | tail -6
timeout 600 python tools/flat_nary_ab.py 2>&1 | grep -v amdgpu | tail -30 | cut -c1-200
