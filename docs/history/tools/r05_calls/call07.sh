#!/bin/bash
# r05 call 7: step account (stamp build), multirank (in-place all-reduce, f64 crossing), round5 tests, 8-rank bench rehearsal over gloo
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/step_account.py > $O/step_account.txt 2>&1; echo "account rc=$?"; cat $O/step_account.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_round5.py tests/test_gpu_shard.py -q -x 2>&1 | tail -12
export SMR_RCCL_LIB=$PWD/tests/libfake_rccl.so
timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu --extras c4 > $O/bench_gpus8_rehearsal.json 2> $O/bench_gpus8_rehearsal.err; echo "rehearsal rc=$?"; cut -c1-600 $O/bench_gpus8_rehearsal.json; tail -5 $O/bench_gpus8_rehearsal.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05/bench_gpus8_rehearsal.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('n_gpus','value','rehearsal')}); print(d['extra'])
except Exception as e: print('ERR',e)
PY
