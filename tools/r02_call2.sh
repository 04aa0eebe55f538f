set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r02
timeout 500 tools/bin/c3_proto 32 64 > gpurun_out/r02/c3_proto2.txt 2>&1; grep -c . gpurun_out/r02/c3_proto2.txt
