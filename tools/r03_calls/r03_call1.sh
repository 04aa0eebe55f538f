# round 3, GPU call 1: where does the time of a 3-5 us launch go (device stamps), non-cubic orbit boxes, fork/join
# step, the power-of-two collapse of the ORBIT kernel (sizes, list grouping, counters), planner cliff.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03a
mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/dev.txt 2>&1
timeout 400 tools/bin/c3_proto2 32 64 128 > $O/c3_proto2.txt 2>&1; echo "proto rc $?"
timeout 200 python tools/device_span.py > $O/device_span.txt 2>&1; echo "span rc $?"; tail -25 $O/device_span.txt
timeout 200 python tools/step_forkjoin.py > $O/step_forkjoin.txt 2>&1; cat $O/step_forkjoin.txt | grep -v amdgpu.ids
timeout 400 python tools/orbit_group_ab.py > $O/orbit_group_ab.txt 2>&1; cat $O/orbit_group_ab.txt | grep -v amdgpu.ids
timeout 500 python tools/cliff_ab.py > $O/cliff_ab.txt 2>&1; grep -c GB/s $O/cliff_ab.txt
bash tools/pmc_orbit_sizes.sh $O/pmc_sum 96,128 > $O/pmc_sum4_96_vs_128.txt 2>&1
PMC_GROUPS="utcl ea_wr chan l1l2" bash tools/pmc_orbit_sizes.sh $O/pmc_perm 96,128 --perm 1 > $O/pmc_perm_96_vs_128.txt 2>&1
rm -rf $O/pmc_sum $O/pmc_perm
grep -E "^n=" $O/c3_proto2.txt | head -80
