#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -8
timeout 600 python tools/gorder_ab.py > $O/gorder_ab2.txt 2>&1; cut -c1-60,200-330 $O/gorder_ab2.txt
timeout 300 python tools/pow2_store_ab.py 120 128 136 > $O/pow2_after.txt 2>&1; cut -c1-120 $O/pow2_after.txt
