set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03m
timeout 600 python tools/flat_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03m/flat_ab.txt
timeout 900 python -m pytest tests/test_gpu_fuzz_families.py -m gpu -q -k "flat or every_family" 2>&1 | tail -15
