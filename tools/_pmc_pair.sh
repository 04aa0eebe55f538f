R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for pair in 0 2; do
  for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/pp; timeout 120 rocprofv3 --pmc $grp -d /tmp/pp -o pp -- python $R/tools/prof_headline.py --which bcast --iters 20 --n 32 --opt orbit_pair=$pair > /tmp/pp.log 2>&1
    echo "== orbit_pair=$pair"; python $R/tools/rocpd_summary.py /tmp/pp/pp_results.db 2>&1 | grep -E "orbit" | cut -c1-110
  done
done
