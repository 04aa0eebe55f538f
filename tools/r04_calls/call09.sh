#!/bin/bash
# r04 call 9: round-4 kernel changes (stepped ranges in TILED, packed STREAM rows, two-level in-launch fold): tests + sanity sweep
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py -q -x 2>&1 | tail -15
timeout 600 python tools/perf_sanity.py > $O/perf_sanity.txt 2>&1; grep -E "step-|257, 129|17, 33, 65|sum \(|100, 90, 80\)" $O/perf_sanity.txt | cut -c1-200
