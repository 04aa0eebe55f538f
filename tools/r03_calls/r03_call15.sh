set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03n
timeout 600 python tools/flat_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03n/flat_ab.txt
timeout 1500 python -m pytest tests/test_gpu_fuzz_families.py -m gpu -q 2>&1 | tail -8
