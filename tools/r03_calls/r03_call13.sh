set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03l
timeout 900 python tools/cliff_ab.py 48 64 96 128 > gpurun_out/r03l/cliff_ab.txt 2>&1; grep "add4\|3 arrays" gpurun_out/r03l/cliff_ab.txt | grep "auto\|run-balanced\|blocks of 4  \|blocks of 4 " | awk -F'|' '{print $1 "|" $2}' | cut -c1-150
