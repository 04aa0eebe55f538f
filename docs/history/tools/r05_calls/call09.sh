#!/bin/bash
# r05 call 9: regenerate the round's evidence files (bench lines, rocprof kernel trace, PMC passes incl. config 5, kernel times, fixed cost, step account, perf sanity)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/round_evidence_r05.sh 2>&1 | tail -60 | cut -c1-250
