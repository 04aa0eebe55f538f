#!/bin/bash
# r04 call 11: STREAM packed rows A/B in one session; round-4 tests
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
cd tools && timeout 600 python stream_pack_ab.py > ../$O/stream_pack_ab.txt 2>&1; cd ..; cat $O/stream_pack_ab.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_round4.py -q -x 2>&1 | tail -3
