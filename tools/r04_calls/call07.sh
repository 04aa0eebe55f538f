#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_replay'], d['config']['step_us_long_graph'])"; done
