set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03g
mkdir -p $O
timeout 900 python tools/orbit_f32.py > $O/orbit_f32.txt 2>&1; grep -v amdgpu.ids $O/orbit_f32.txt
