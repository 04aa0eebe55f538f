#!/bin/bash
# r04 call 15: GPU suite on 4 workers with failure names; rocprofv3 of the bench command
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -rf -n 4 -p no:cacheprovider 2>&1 | tail -40 | cut -c1-300 | tee $O/pytest_gpu.txt
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1000 --no-cpu --no-extra > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1; echo "rocprof rc=$?"
grep -v "^W2026\|^I2026\|^E2026" $GRAFT_REPO_ROOT/$O/prof_bench.log | tail -3 | cut -c1-900
