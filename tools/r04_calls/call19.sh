#!/bin/bash
# r04 call 19: multi-rank test (4 ranks on library-owned streams), eager + seq tests, bench
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_eager.py tests/test_gpu_seq.py tests/test_gpu_round4.py -q -rf 2>&1 | tail -8
timeout 600 python bench.py --no-cpu --no-extra > $O/bench_noextra.json 2>$O/bench_noextra.err; python -c "
import json; d=json.load(open('$O/bench_noextra.json')); print(d['value'], d['ms_per_step'], d['config']['step_us_long_graph'])"
