#!/bin/bash
# r05 call 1: fence matrix of the replayed bench step (acquire by need / release scope / asymmetric slices) + seq tests + baseline bench line
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/fence_ab.py > $O/fence_ab.txt 2>&1; echo "fence_ab rc=$?"; cat $O/fence_ab.txt
timeout 400 python -m pytest tests/test_gpu_seq.py -q -x 2>&1 | tail -4
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_k20_a.json 2> $O/bench_k20_a.err; echo "bench rc=$?"; cut -c1-400 $O/bench_k20_a.json
