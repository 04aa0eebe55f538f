#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 400 python tools/pow2_sweep.py 128 sum > $O/pow2_sweep_sum128.txt 2>&1; cat $O/pow2_sweep_sum128.txt | cut -c1-220
timeout 300 python tools/pow2_sweep.py 128 perm > $O/pow2_sweep_perm128.txt 2>&1; cat $O/pow2_sweep_perm128.txt | cut -c1-220
timeout 300 python tools/pow2_sweep.py 136 perm > $O/pow2_sweep_perm136.txt 2>&1; cat $O/pow2_sweep_perm136.txt | cut -c1-220
