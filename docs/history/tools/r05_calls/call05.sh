#!/bin/bash
# r05 call 5: self-released launches (write-through stores, no release fence): tests, hazard demo, store-mode matrix on the product build, bench
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_seq.py tests/test_gpu_eager.py -q 2>&1 | tail -15
timeout 300 python tools/waw_hazard.py > $O/waw_hazard.txt 2>&1; echo "hazard rc=$?"; cat $O/waw_hazard.txt
timeout 300 python tools/store_mode_ab.py > $O/store_mode_ab.txt 2>&1; echo "store rc=$?"; cut -c1-330 $O/store_mode_ab.txt
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_k20_d.json 2> $O/bench_k20_d.err; echo "bench rc=$?"; cut -c1-300 $O/bench_k20_d.json; tail -3 $O/bench_k20_d.err
