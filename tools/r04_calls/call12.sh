#!/bin/bash
# r04 call 12: block-range slices of single-launch components
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_seq.py -q -x 2>&1 | tail -8
timeout 180 tools/bin/overlap_probe strided.jl_amd/libstrided_hip.so 32 500 > $O/overlap_product3.txt 2>&1; echo "rc=$?"; grep -A1 "seq AQL\|graph, in order" $O/overlap_product3.txt | cut -c1-330
