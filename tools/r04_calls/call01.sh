#!/bin/bash
# r04 call 1: overlap probe (stamp + product builds), does the --offload-compress library load and pass parity
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 120 tools/bin/overlap_probe strided.jl_amd/libstrided_hip_stamp.so 32 200 > $O/overlap_stamp.txt 2>&1; tail -30 $O/overlap_stamp.txt
timeout 120 tools/bin/overlap_probe strided.jl_amd/libstrided_hip.so 32 500 > $O/overlap_product.txt 2>&1; tail -14 $O/overlap_product.txt
GPU_MAX_HW_QUEUES=8 timeout 120 tools/bin/overlap_probe strided.jl_amd/libstrided_hip.so 32 500 > $O/overlap_product_q8.txt 2>&1; tail -14 $O/overlap_product_q8.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
