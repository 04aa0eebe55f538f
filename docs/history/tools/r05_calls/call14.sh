#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/eager_self_release.py > $O/eager_self_release.txt 2>&1; cat $O/eager_self_release.txt
timeout 300 python -m pytest tests/test_gpu_eager.py -q 2>&1 | tail -3
