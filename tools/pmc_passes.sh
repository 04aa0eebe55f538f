#!/bin/bash
# PMC passes (one counter group per run, no tracing domains besides the kernel trace) for one workload of
# tools/prof_headline.py.  Usage on the GPU box: bash tools/pmc_passes.sh <outdir> [which] [n] [iters]
set -u
OUT=${1:-gpurun_out/prof}; WHICH=${2:-both}; NN=${3:-32}; ITERS=${4:-20}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
run() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" -d $R/$OUT/$name -o $name -- python $R/tools/prof_headline.py --which $WHICH --iters $ITERS --n $NN > $R/$OUT/$name.log 2>&1; }
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run tcc2 TCC_EA0_WRREQ_sum TCC_WRITE_sum TCC_READ_sum TCC_TAG_STALL_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY
run valu SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
for d in tcp tcc1 tcc2 lds sq valu fetch write; do
  [ -f $OUT/$d/${d}_results.db ] && python tools/rocpd_summary.py $OUT/$d/${d}_results.db | grep -v at6native | grep -vE "^ +[0-9]+ +[0-9]+ .*kernel$" || { echo "== $d: no database"; tail -3 $OUT/$d.log; }
done
