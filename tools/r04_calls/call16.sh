#!/bin/bash
# r04 call 16: eager direct dispatch on library-owned streams
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_eager.py tests/test_gpu_seq.py -q -x 2>&1 | tail -15
timeout 300 python tools/eager_host_cost.py > $O/eager_host_cost.txt 2>&1; cat $O/eager_host_cost.txt
