#!/bin/bash
# r04 call 4: AQL replay with one hardware queue per dependency component
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 180 tools/bin/overlap_probe strided.jl_amd/libstrided_hip_stamp.so 32 200 > $O/overlap_stamp2.txt 2>&1; echo "rc=$?"; grep -A1 "seq AQL" $O/overlap_stamp2.txt
timeout 180 tools/bin/overlap_probe strided.jl_amd/libstrided_hip.so 32 500 > $O/overlap_product2.txt 2>&1; echo "rc=$?"; grep -A1 "seq AQL\|eager" $O/overlap_product2.txt
