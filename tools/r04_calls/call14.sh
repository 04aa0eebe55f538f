#!/bin/bash
# r04 call 14: whole GPU suite, default bench, rocprofv3 kernel trace of the bench command (sequences replay through HIP under the profiler)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_k20.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_k20.json')); print('K=20:', d['value'], d['ms_per_step'], d['ms_per_step_replay']); print(d['roofline']['independent_launches'])"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1000 --no-cpu --no-extra > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1; echo "rocprof rc=$?"
tail -2 $GRAFT_REPO_ROOT/$O/prof_bench.log | cut -c1-600
find $GRAFT_REPO_ROOT/$O/prof_bench -type f | head
