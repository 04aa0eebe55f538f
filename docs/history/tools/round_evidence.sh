#!/bin/bash
# Regenerates the evidence files of a round on the GPU box (profiles/README.md says what each one is).
# Usage: bash tools/round_evidence.sh r02      -> gpurun_out/<tag>/..., copied to profiles/ afterwards
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json; echo
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python $R/bench.py --steps 1000 --no-cpu --no-extra > $R/$O/kt.log 2>&1 )
python tools/rocpd_summary.py --hist $O/kt/kt_results.db > $O/bench_kernel_trace_stats.txt 2>&1; head -12 $O/bench_kernel_trace_stats.txt
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $R/$O/gt -o gt -- python $R/tools/graph_trace.py > $R/$O/graph_trace.log 2>&1 )
( grep -E "^(perm|bcast)" $O/graph_trace.log; python tools/rocpd_summary.py --hist $O/gt/gt_results.db ) > $O/graph_trace_hist.txt 2>&1
python tools/graph_trace.py 2>/dev/null | grep -E "^(perm|bcast):" | sed 's/under rocprofv3 these include the profiler.s per-dispatch overhead/NOT profiled/' > $O/graph_trace_unprofiled.txt; cat $O/graph_trace_unprofiled.txt
bash tools/pmc_passes.sh $O/pmc32 both 32 20 > $O/pmc_headline_kernels.txt 2>&1; grep -E "FETCH_SIZE|WRITE_SIZE|TCP_TCC_READ_REQ" $O/pmc_headline_kernels.txt | head
bash tools/pmc_passes.sh $O/pmc_c1 c1 32 10 > $O/pmc_c1_symmetrise_4000.txt 2>&1
bash tools/pmc_passes.sh $O/pmc_c4 c4 32 5 > $O/pmc_c4_mapreduce_abs2.txt 2>&1
bash tools/pmc_passes.sh $O/pmc_c5 c5 32 5 > $O/pmc_c5_expr_8192.txt 2>&1
bash tools/pmc_passes.sh $O/pmc128 bcast 128 3 > $O/pmc_bcast4_128.txt 2>&1
g++ -O2 tools/host_overhead.cpp -Iinclude -Lstrided.jl_amd -lstrided_hip -Wl,-rpath,$R/strided.jl_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o /tmp/host_overhead 2> $O/host_overhead.txt && /tmp/host_overhead >> $O/host_overhead.txt 2>&1
python tools/host_overhead.py 2>/dev/null >> $O/host_overhead.txt; cat $O/host_overhead.txt
python tools/orbit_sweep.py 2>/dev/null > $O/orbit_sweep.txt; tail -3 $O/orbit_sweep.txt
python tools/orbit_ab.py 2>/dev/null > $O/orbit_ab.txt
python tools/perm_ab.py 2>/dev/null > $O/perm_sizes.txt
rm -rf $O/kt $O/gt $O/pmc32 $O/pmc_c1 $O/pmc_c4 $O/pmc_c5 $O/pmc128
ls -la $O
