# round 3, GPU call 2: read side / write side of the orbit access pattern over sizes around 128 (any multiple of 8) and
# list orders; per-size counters; persistent pipelined orbits at cache-resident sizes; block tile order for distinct arrays
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03b
mkdir -p $O
timeout 600 tools/bin/c3_proto2 sweep 64 96 112 120 128 136 144 > $O/c3_proto2_sweep.txt 2>&1; echo "proto rc $?"; grep "^n=" $O/c3_proto2_sweep.txt | cut -c1-200
timeout 200 python tools/device_span.py > $O/device_span.txt 2>&1; echo "span rc $?"; grep -v amdgpu.ids $O/device_span.txt
timeout 300 python tools/orbit_pipe32.py > $O/orbit_pipe32.txt 2>&1; grep -v amdgpu.ids $O/orbit_pipe32.txt
timeout 600 python tools/cliff_ab.py 48 64 96 128 > $O/cliff_ab.txt 2>&1; grep "add4\|3 arrays" $O/cliff_ab.txt | awk -F'|' '{print $1 "|" $2}' | cut -c1-160
for n in 96 128 144; do
  PMC_GROUPS="ea_rd ea_wr evict dram l1l2" bash tools/pmc_orbit_sizes.sh $O/pmc_sum_$n $n > $O/pmc_sum4_$n.txt 2>&1
  rm -rf $O/pmc_sum_$n
done
