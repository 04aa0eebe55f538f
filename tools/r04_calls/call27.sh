#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_eager.py tests/test_gpu_seq.py tests/test_gpu_multirank.py tests/test_gpu_round4.py tests/test_gpu_flatb.py tests/test_gpu_shard.py -q -rf -n 3 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
