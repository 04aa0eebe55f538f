#!/bin/bash
# Regenerates the round-6 evidence files on the GPU box (profiles/README.md says what each one is).
# Usage: bash tools/round_evidence_r06.sh [part ...]   parts: bench prof probe times sanity rehearsal tests (default: all but tests)
# Writes gpurun_out/r06e/...; copied to profiles/r06_* afterwards.  Needs tools/bin/orbit32_probe{,_preload} and tools/bin/orbit16_probe (built on the CPU box:
#   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 [-mllvm -amdgpu-kernarg-preload-count=16] tools/orbit32_probe.hip -o tools/bin/...)
# and strided.jl_amd/libstrided_hip_stamp.so (make -C strided.jl_amd/csrc stamp) for the device-stamp parts.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06e
mkdir -p $O
PARTS=${*:-bench prof probe times sanity rehearsal}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then
  python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 200 $O/bench_n1.json; echo
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_style.json 2>/dev/null; cut -c1-200 $O/bench_n1_driver_style.json
fi
if has prof; then
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python $R/bench.py --steps 1000 --no-cpu --no-extra > $R/$O/kt.log 2>&1 ); echo "rocprof rc=$?"
  python tools/rocpd_summary.py --hist $O/kt/kt_results.db > $O/bench_kernel_trace_stats.txt 2>&1; head -5 $O/bench_kernel_trace_stats.txt | cut -c1-160
  bash tools/pmc_passes.sh $O/pmc32 both 32 20 > $O/pmc_headline_kernels.txt 2>&1; grep -E "FETCH_SIZE|WRITE_SIZE|TCP_TCC_READ_REQ" $O/pmc_headline_kernels.txt | head -8 | cut -c1-200
  rm -rf $O/kt $O/pmc32
fi
if has probe; then
  timeout 300 tools/bin/orbit32_probe > $O/orbit32_probe.txt 2>&1; echo "probe rc=$?"
  timeout 300 tools/bin/orbit32_probe_preload > $O/orbit32_probe_preload.txt 2>&1; echo "probe (preload build) rc=$?"
  timeout 300 tools/bin/orbit16_probe > $O/orbit16_probe.txt 2>&1; echo "orbit16_probe rc=$?"
  timeout 600 python tools/orbit_pack_ab.py > $O/orbit_pack_ab.txt 2>&1
  timeout 600 python tools/cold_orbit_sweep.py 32 48 64 > $O/cold_orbit_sweep.txt 2>&1
fi
if has times; then
  timeout 300 python tools/kernel_times.py --rocprof $O/bench_kernel_trace_stats.txt > $O/kernel_times.txt 2>&1; head -24 $O/kernel_times.txt | cut -c1-200
  timeout 300 python tools/seq_fixed_cost.py 2>/dev/null > $O/seq_fixed_cost.txt
  timeout 300 python tools/step_account.py > $O/step_account.txt 2>&1
fi
if has sanity; then
  timeout 600 python tools/perf_sanity.py > $O/perf_sanity.txt 2>/dev/null; head -3 $O/perf_sanity.txt | cut -c1-160
  timeout 300 python tools/sum_cases.py > $O/sum_cases.txt 2>/dev/null
  timeout 300 python tools/sum_cases.py reduce_col_exact=0 > $O/sum_cases_exact0.txt 2>/dev/null
  timeout 300 python tools/ragged_cases.py > $O/ragged_tiles.txt 2>/dev/null
  timeout 300 python tools/ragged_family_ab.py > $O/ragged_family_ab.txt 2>/dev/null
  timeout 300 python tools/row_stride_cases.py > $O/row_stride_cases.txt 2>/dev/null
fi
if has rehearsal; then
  for n in 2 8; do
    SMR_RCCL_LIB=tests/libfake_rccl.so timeout 900 python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu --extras c4,sharded > $O/bench_gpus${n}_rehearsal.json 2> $O/bench_gpus${n}.err; echo "rehearsal $n rc=$?"
  done
fi
if has tests; then
  timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
fi
ls $O
