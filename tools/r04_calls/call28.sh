#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_eager.py -q -rf 2>&1 | tail -3
