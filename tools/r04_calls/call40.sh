#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python -m pytest tests/test_gpu_eager.py tests/test_gpu_seq.py tests/test_gpu_round4.py -q -n 3 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
timeout 120 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-220
