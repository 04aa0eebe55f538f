#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 600 python tools/gorder_ab.py > $O/gorder_ab.txt 2>&1; cat $O/gorder_ab.txt | cut -c1-330
