set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03p
timeout 600 python tools/flat_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03p/flat_ab.txt | grep "(0, 2, 1)\|(0, 3, 2, 1)"
timeout 1200 python -m pytest tests/test_gpu_fuzz_families.py -m gpu -q -k "flat or tiled_short0 or tiled-" 2>&1 | tail -4
