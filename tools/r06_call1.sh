#!/bin/bash
# round 6, evidence call 1: probes (pattern ceilings / variants / phases) and the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06d
mkdir -p $O
timeout 300 tools/bin/orbit32_probe > $O/orbit32_probe.txt 2>&1; echo "probe rc=$?"
timeout 300 tools/bin/orbit32_probe_preload > $O/orbit32_probe_preload.txt 2>&1; echo "probe_preload rc=$?"
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.txt
