#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/seq_plan_ab.py > $O/seq_plan_ab.txt 2>&1; cat $O/seq_plan_ab.txt | cut -c1-300
