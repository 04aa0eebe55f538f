#!/bin/bash
# Regenerates the round-3 evidence files on the GPU box (profiles/README.md says what each one is).
# Usage: bash tools/round_evidence_r03.sh   -> gpurun_out/r03/..., copied to profiles/r03_* afterwards
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03
mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_n1_driver_style.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python $R/bench.py --steps 1000 --no-cpu --no-extra > $R/$O/kt.log 2>&1 )
python tools/rocpd_summary.py --hist $O/kt/kt_results.db > $O/bench_kernel_trace_stats.txt 2>&1; head -8 $O/bench_kernel_trace_stats.txt
timeout 200 python tools/device_span.py > $O/device_span.txt 2>&1; grep -v amdgpu.ids $O/device_span.txt | tail -20
bash tools/pmc_passes.sh $O/pmc32 both 32 20 > $O/pmc_headline_kernels.txt 2>&1; grep -E "FETCH_SIZE|WRITE_SIZE|TCP_TCC_READ_REQ" $O/pmc_headline_kernels.txt | head
for n in 96 128 144; do
  bash tools/pmc_orbit_sizes.sh $O/pmc_sum_$n $n > $O/pmc_sum4_$n.txt 2>&1
  rm -rf $O/pmc_sum_$n
done
timeout 400 tools/bin/c3_proto2 32 > $O/c3_proto2_32.txt 2>&1
timeout 600 tools/bin/c3_proto2 sweep 64 96 112 120 128 136 144 > $O/c3_proto2_sweep.txt 2>&1
timeout 200 python tools/step_forkjoin.py > $O/step_forkjoin.txt 2>&1
timeout 300 python tools/orbit_pipe32.py > $O/orbit_pipe32.txt 2>&1
timeout 400 python tools/orbit_group_ab.py > $O/orbit_group_ab.txt 2>&1
timeout 900 python tools/cliff_ab.py 32 48 64 96 128 > $O/cliff_ab.txt 2>&1
timeout 600 python tools/perm_block_ab.py > $O/perm_block_ab.txt 2>&1
timeout 600 python tools/orbit_f32.py > $O/orbit_f32.txt 2>&1
timeout 600 python tools/orbit_cplx.py > $O/orbit_cplx.txt 2>&1
timeout 300 python tools/orbit_sweep.py 2>/dev/null > $O/orbit_sweep.txt
g++ -O2 tools/host_overhead.cpp -Iinclude -Lstrided.jl_amd -lstrided_hip -Wl,-rpath,$R/strided.jl_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o /tmp/host_overhead 2> $O/host_overhead.txt && /tmp/host_overhead >> $O/host_overhead.txt 2>&1
python tools/host_overhead.py 2>/dev/null >> $O/host_overhead.txt; cat $O/host_overhead.txt
timeout 600 python tools/perf_sanity.py > $O/perf_sanity.txt 2>/dev/null; head -5 $O/perf_sanity.txt
rm -rf $O/kt $O/pmc32
ls -la $O
