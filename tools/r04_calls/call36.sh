#!/bin/bash
# r04 call 36: final state -- the whole GPU suite in one process, then the evidence files
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu -rf -p no:cacheprovider > $O/pytest_gpu_full.txt 2>&1; tail -2 $O/pytest_gpu_full.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
bash tools/round_evidence_r04.sh 2>&1 | tail -12 | cut -c1-200
