#!/bin/bash
# r04 call 2: two host threads on two streams
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 120 tools/bin/overlap_probe strided.jl_amd/libstrided_hip.so 32 500 > $O/overlap_product2.txt 2>&1; tail -14 $O/overlap_product2.txt
