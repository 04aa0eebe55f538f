#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/cold_seq_ab.py > $O/cold_seq_ab.txt 2>&1; cat $O/cold_seq_ab.txt | cut -c1-230
