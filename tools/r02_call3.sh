set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r02
timeout 600 python tools/orbit_ab.py > gpurun_out/r02/orbit_ab.txt 2>&1; cat gpurun_out/r02/orbit_ab.txt | grep -v amdgpu.ids
