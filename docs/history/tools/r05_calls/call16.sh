#!/bin/bash
# r05 call 16: the whole GPU suite in ONE process, as the driver runs it, + smoke + the driver's bench form
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/pytest_gpu_full.txt; cat $O/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20_e.json 2> $O/bench_k20_e.err; echo "bench rc=$?"; cut -c1-260 $O/bench_k20_e.json
