#!/bin/bash
# Regenerates the round-5 evidence files on the GPU box (profiles/README.md says what each one is).
# Usage: bash tools/round_evidence_r05.sh   -> gpurun_out/r05e/..., copied to profiles/r05_* afterwards
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05e
mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 200 $O/bench_n1.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_style.json 2>/dev/null; cut -c1-200 $O/bench_n1_driver_style.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python $R/bench.py --steps 1000 --no-cpu --no-extra > $R/$O/kt.log 2>&1 ); echo "rocprof rc=$?"
python tools/rocpd_summary.py --hist $O/kt/kt_results.db > $O/bench_kernel_trace_stats.txt 2>&1; head -5 $O/bench_kernel_trace_stats.txt | cut -c1-160
bash tools/pmc_passes.sh $O/pmc32 both 32 20 > $O/pmc_headline_kernels.txt 2>&1; grep -E "FETCH_SIZE|WRITE_SIZE|TCP_TCC_READ_REQ" $O/pmc_headline_kernels.txt | head -8 | cut -c1-200
bash tools/pmc_passes.sh $O/pmcc5 c5 8192 5 > $O/pmc_c5_expr_8192.txt 2>&1; grep -E "SQ_ACTIVE_INST_VALU|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES" $O/pmc_c5_expr_8192.txt | head -8 | cut -c1-200
timeout 300 python tools/kernel_times.py --rocprof $O/bench_kernel_trace_stats.txt > $O/kernel_times.txt 2>&1; head -24 $O/kernel_times.txt | cut -c1-200
timeout 300 python tools/seq_fixed_cost.py 2>/dev/null > $O/seq_fixed_cost.txt
timeout 300 python tools/step_account.py > $O/step_account.txt 2>&1
timeout 600 python tools/perf_sanity.py > $O/perf_sanity.txt 2>/dev/null; head -3 $O/perf_sanity.txt | cut -c1-160
rm -rf $O/kt $O/pmc32 $O/pmcc5
ls $O
