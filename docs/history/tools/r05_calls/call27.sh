#!/bin/bash
# round 5, call 27: bench.py's N-rank path with the watchdog around the secondary workloads -- normal case (8 ranks, rehearsal) and a
# simulated stuck rank (2 ranks: rank 1 never enters the legs, rank 0 blocks in the sharded config 4's collective)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
export SMR_RCCL_LIB=$PWD/tests/libfake_rccl.so
timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu --extras c4 2>/dev/null | tail -1 | cut -c1-300; echo "rc=$?"
BENCH_SIMULATE_STUCK_RANK=1 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --extras c4 --extra-timeout 20 > gpurun_out/r05/stuck.json 2> gpurun_out/r05/stuck.err; echo "stuck-case rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/stuck.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['extra'])
PY
