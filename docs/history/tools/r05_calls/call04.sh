#!/bin/bash
# r05 call 4: store policy at HBM sizes (pow2 collapse), product vs sc1 build
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 600 python tools/pow2_store_ab.py > $O/pow2_store_ab.txt 2>&1; echo "rc=$?"; cat $O/pow2_store_ab.txt
