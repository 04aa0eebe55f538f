#!/bin/bash
# r05 call 3: round-5 dispatch tests again + first-acquire by memory type + store-mode experiment (sc1 write-through build)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_seq.py tests/test_gpu_eager.py -q 2>&1 | tail -25
timeout 300 python tools/store_mode_ab.py > $O/store_mode_ab.txt 2>&1; echo "store rc=$?"; cat $O/store_mode_ab.txt
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_k20_c.json 2> $O/bench_k20_c.err; echo "bench rc=$?"; cut -c1-300 $O/bench_k20_c.json; tail -3 $O/bench_k20_c.err
