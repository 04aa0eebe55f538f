R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06c
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/ktp -o kt -- $R/tools/bin/orbit32_probe_preload quick > $R/$O/ktp.log 2>&1 ); echo "rc=$?"
python tools/rocpd_summary.py $O/ktp/kt_results.db > $O/kt_probe.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/ktl -o kt -- python $R/tools/orbit_pack_ab.py "32^4 f64" > $R/$O/ktl.log 2>&1 ); echo "rc=$?"
python tools/rocpd_summary.py $O/ktl/kt_results.db > $O/kt_lib.txt 2>&1
rm -rf $O/ktp $O/ktl
