#!/bin/bash
# r04 call 5: sequence tests + bench with the seq step mode
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_seq.py -q -x 2>&1 | tail -15
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_seq20.txt 2>&1; tail -1 $O/bench_seq20.txt | cut -c1-1800
timeout 600 python bench.py --no-cpu --no-extra > $O/bench_seq.txt 2>&1; tail -1 $O/bench_seq.txt | cut -c1-900
