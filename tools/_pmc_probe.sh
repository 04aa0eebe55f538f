R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pq; timeout 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d /tmp/pq -o pq -- $R/tools/bin/orbit16_probe > /tmp/pq.log 2>&1
python $R/tools/rocpd_summary.py /tmp/pq/pq_results.db 2>&1 | grep -E "k_sum16" | cut -c1-100
