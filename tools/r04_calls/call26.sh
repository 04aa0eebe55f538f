#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python bench.py --no-cpu > $O/bench_b.json 2>$O/bench_b.err; python -c "
import json; d=json.load(open('$O/bench_b.json')); print(d['value'], d['ms_per_step']); print({k:(v.get('us'),v.get('GB/s',v.get('GB/s_total'))) for k,v in d['extra'].items()})"; tail -3 $O/bench_b.err
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=20', d['value'], d['ms_per_step'], d['ms_per_step_replay'])"; done
