#!/bin/bash
# r04 call 18: eager direct dispatch with resident argument blocks: tests, host cost from C and from Python
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_eager.py -q -x 2>&1 | tail -5
timeout 120 tools/bin/host_overhead > $O/host_overhead.txt 2>&1; cat $O/host_overhead.txt
timeout 300 python tools/eager_host_cost.py > $O/eager_host_cost.txt 2>&1; cat $O/eager_host_cost.txt | grep -v amdgpu.ids
