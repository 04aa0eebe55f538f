#!/bin/bash
# r04 call 10: in-launch tree vs second launch; STREAM forms after the compile-time split
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/reduce_tree_ab.py > $O/reduce_tree_ab.txt 2>&1; cat $O/reduce_tree_ab.txt | cut -c1-220
timeout 600 python tools/perf_sanity.py > $O/perf_sanity2.txt 2>&1; grep -E "257, 129|17, 33, 65|100, 90, 80\)|sub-box|step-" $O/perf_sanity2.txt | cut -c1-180
timeout 600 python -m pytest tests/test_gpu_round4.py -q -x 2>&1 | tail -3
