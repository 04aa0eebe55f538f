set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -4
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; tail -c 3000 gpurun_out/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt -- python $R/bench.py --steps 1000 --no-cpu --no-extra > $R/gpurun_out/prof/kt.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof/kt/kt_results.db > gpurun_out/prof/kt_summary.txt 2>&1; head -30 gpurun_out/prof/kt_summary.txt
bash tools/pmc_passes.sh gpurun_out/prof both > gpurun_out/prof/pmc_summary.txt 2>&1; tail -60 gpurun_out/prof/pmc_summary.txt
