set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03c
mkdir -p $O
timeout 300 tools/bin/c3_proto2 32 > $O/c3_proto2_32.txt 2>&1; grep "^n= 32 orbit" $O/c3_proto2_32.txt | cut -c1-150
timeout 900 python tools/cliff_ab.py 32 48 64 96 128 > $O/cliff_ab.txt 2>&1; grep "add4\|3 arrays" $O/cliff_ab.txt | awk -F'|' '{print $1 "|" $2 "|" $3}' | cut -c1-230
