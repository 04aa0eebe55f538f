#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_seq.py tests/test_gpu_eager.py tests/test_seq_components.py -q -n 2 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=20', d['value'], d['ms_per_step'])"
