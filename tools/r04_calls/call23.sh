#!/bin/bash
# r04 call 23: the whole GPU suite the way the driver runs it
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu -rf -p no:cacheprovider 2>&1 | tail -12 | cut -c1-400 | tee $O/pytest_gpu.txt
