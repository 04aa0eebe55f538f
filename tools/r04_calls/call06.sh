#!/bin/bash
# r04 call 6: sequence tests (JIT kernels as AQL packets) + fixed cost of short replays
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_seq.py tests/test_jit.py -q 2>&1 | tail -15
timeout 300 python tools/seq_fixed_cost.py > $O/seq_fixed_cost.txt 2>&1; cat $O/seq_fixed_cost.txt
