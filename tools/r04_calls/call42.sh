#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 48 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_shard.py -q -x 2>&1 | tail -3
