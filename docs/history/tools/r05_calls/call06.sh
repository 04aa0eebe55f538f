#!/bin/bash
# r05 call 6: the whole GPU suite + smoke with the new dispatch defaults
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-200
