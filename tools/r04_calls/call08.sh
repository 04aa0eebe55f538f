#!/bin/bash
# r04 call 8: smr_comm.cpp with 2 / 4 / 8 processes on one GPU (fake RCCL over shared memory)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_multirank.py -q -x 2>&1 | tail -40 | tee $O/multirank.txt
