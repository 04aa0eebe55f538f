set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03d
mkdir -p $O
timeout 900 python tools/perm_block_ab.py > $O/perm_block_ab.txt 2>&1; grep -v amdgpu.ids $O/perm_block_ab.txt
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
