set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python -c "import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_kernel']['permutedims']['us'], d['roofline']['per_kernel']['broadcast4']['us'])"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_n1_driver_style.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python $R/bench.py --steps 1000 --no-cpu --no-extra > $R/$O/kt.log 2>&1 )
python tools/rocpd_summary.py --hist $O/kt/kt_results.db > $O/bench_kernel_trace_stats.txt 2>&1; head -5 $O/bench_kernel_trace_stats.txt | cut -c1-150
timeout 200 python tools/device_span.py > $O/device_span.txt 2>&1; grep "cadence" $O/device_span.txt
timeout 600 python tools/perf_sanity.py > $O/perf_sanity.txt 2>/dev/null; head -12 $O/perf_sanity.txt | cut -c1-160
rm -rf $O/kt
