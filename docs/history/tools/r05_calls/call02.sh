#!/bin/bash
# r05 call 2: hardened direct dispatch (metadata layout + self-test, async run, per-stream pending, failure path) + fixed cost + bench
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_seq.py tests/test_gpu_eager.py -q -x 2>&1 | tail -25
timeout 200 python tools/seq_fixed_cost.py > $O/seq_fixed_cost.txt 2>&1; echo "fixed rc=$?"; cut -c1-260 $O/seq_fixed_cost.txt | grep -v "K=  *\(2\|5\|50\|100\|500\) "
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_k20_b.json 2> $O/bench_k20_b.err; echo "bench rc=$?"; cut -c1-300 $O/bench_k20_b.json; tail -3 $O/bench_k20_b.err
