#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 300 python tools/seq_vs_eager.py > $O/seq_vs_eager.txt 2>&1; grep -v amdgpu.ids $O/seq_vs_eager.txt | cut -c1-220
