#!/bin/bash
# round 3: final validation on the GPU box -- the whole GPU suite, smoke, the two bench invocations, kernel-trace stats, planner sweep
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_style.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/bench_n1_driver_style.json').read().strip().splitlines()[-1]);print('driver-style', d['value'], d['ms_per_step'], d['roofline'])"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python $R/bench.py --steps 1000 --no-cpu --no-extra > $R/$O/kt.log 2>&1 )
python tools/rocpd_summary.py --hist $O/kt/kt_results.db > $O/bench_kernel_trace_stats.txt 2>&1; head -8 $O/bench_kernel_trace_stats.txt
rm -rf $O/kt
timeout 600 python tools/perf_sanity.py > $O/perf_sanity.txt 2>/dev/null; head -3 $O/perf_sanity.txt
