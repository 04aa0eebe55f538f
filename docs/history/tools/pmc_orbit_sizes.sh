#!/bin/bash
# PMC passes that compare the ORBIT kernel (4-way permuted sum) at a power-of-two size with a neighbouring size:
# address translation (UTCL1), memory-side request sizes, write-backs, DRAM credit stalls, per-channel spread.
# One counter group per run (no tracing domains besides the kernel trace).
# Usage on the GPU box: bash tools/pmc_orbit_sizes.sh <outdir> [sizes] [extra args of prof_orbit_sizes.py]
set -u
OUT=${1:-gpurun_out/pmc_sizes}; SIZES=${2:-96,128}; shift; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
GROUPS_=${PMC_GROUPS:-"utcl utcl2 ea_rd ea_wr evict dram chan l1l2"}
run() { name=$1; shift; case " $GROUPS_ " in *" $name "*) ;; *) return;; esac; timeout 150 rocprofv3 --pmc "$@" -d $R/$OUT/$name -o $name -- python $R/tools/prof_orbit_sizes.py --sizes $SIZES --iters 3 $EXTRA > $R/$OUT/$name.log 2>&1; }
EXTRA="$*"
run utcl  TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_THRASHING_STALL_sum
run utcl2 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
run ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run evict TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_HIT_sum TCC_MISS_sum
run dram  TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
run chan  TCC_EA0_RDREQ TCC_EA0_WRREQ
run l1l2  TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
cd $R
for d in $GROUPS_; do
  [ -f $OUT/$d/${d}_results.db ] && python tools/rocpd_summary.py --by-grid $OUT/$d/${d}_results.db | grep -v at6native | grep -vE "^ +[0-9]+ +[0-9]+ .*kernel$" || { echo "== $d: no database"; tail -3 $OUT/$d.log; }
done
