#!/bin/bash
# r04 call 3 (re-entry; the outputs of calls 1-2 were lost with the container): overlap probe, stamp + product builds, incl. the AQL sequences
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 180 tools/bin/overlap_probe strided.jl_amd/libstrided_hip_stamp.so 32 200 > $O/overlap_stamp.txt 2>&1; echo "rc=$?"; tail -40 $O/overlap_stamp.txt
timeout 180 tools/bin/overlap_probe strided.jl_amd/libstrided_hip.so 32 500 > $O/overlap_product.txt 2>&1; echo "rc=$?"; tail -30 $O/overlap_product.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_short.txt 2>&1; tail -2 $O/bench_short.txt | cut -c1-1500
