#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
for n in 128 96 120 136 64; do timeout 300 python tools/pow2_sweep.py $n sum short 2>&1 | grep -v amdgpu.ids | cut -c1-100; done
