#!/bin/bash
# r04 call 17: eager direct dispatch: tests; host cost with device / host argument blocks and GPU-only / interrupt signals
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_eager.py -q -x 2>&1 | tail -12
echo "== default"; timeout 300 python tools/eager_host_cost.py > $O/eager_host_cost.txt 2>&1; cat $O/eager_host_cost.txt | grep -v amdgpu.ids
echo "== SMR_EAGER_KERNARG=host"; SMR_EAGER_KERNARG=host timeout 300 python tools/eager_host_cost.py 2>&1 | grep "library stream\|\^4\|{" 
echo "== SMR_EAGER_SIGNALS=interrupt"; SMR_EAGER_SIGNALS=interrupt timeout 300 python tools/eager_host_cost.py 2>&1 | grep "library stream\|\^4\|{"
