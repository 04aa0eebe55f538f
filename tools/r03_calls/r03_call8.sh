set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03h
mkdir -p $O
timeout 900 python tools/orbit_cplx.py > $O/orbit_cplx.txt 2>&1; grep -v amdgpu.ids $O/orbit_cplx.txt
timeout 200 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; grep -v amdgpu.ids $O/host_overhead.txt
