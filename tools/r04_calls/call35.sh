#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_fuzz_families.py -q -k "pairs or ragged or nary or flat" -n 2 2>&1 | tail -3
