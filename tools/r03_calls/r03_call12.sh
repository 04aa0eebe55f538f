set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03k
for dt in f32 c128; do DT=$dt timeout 600 python tools/perm_block_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-40,41-75,200-260 ; done | tee gpurun_out/r03k/perm_other_types.txt
