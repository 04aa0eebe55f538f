set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03i
timeout 900 python tools/orbit_skew_ab.py > gpurun_out/r03i/orbit_skew_ab.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03i/orbit_skew_ab.txt
