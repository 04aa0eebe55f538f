set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r03/pytest_gpu.txt | tail -5
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/round_evidence_r03.sh > gpurun_out/r03/evidence.log 2>&1; tail -40 gpurun_out/r03/evidence.log
