#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_fuzz_families.py -q -k "flat_batched or flat2_long" -s 2>&1 | grep -E "fuzz families|passed|failed|Error|error|assert|^E " | head -30 | cut -c1-300
