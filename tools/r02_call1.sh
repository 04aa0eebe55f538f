set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r02
echo "== c3_proto"; timeout 400 tools/bin/c3_proto 32 64 128 > gpurun_out/r02/c3_proto.txt 2>&1; tail -n 100 gpurun_out/r02/c3_proto.txt
echo "== pow2_probe"; timeout 300 python tools/pow2_probe.py > gpurun_out/r02/pow2_probe.txt 2>&1; cat gpurun_out/r02/pow2_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/r02/kt -o kt -- python $R/tools/graph_trace.py > $R/gpurun_out/r02/graph_trace.log 2>&1
cd $R
tail -5 gpurun_out/r02/graph_trace.log
python tools/rocpd_summary.py --hist gpurun_out/r02/kt/kt_results.db > gpurun_out/r02/graph_trace_hist.txt 2>&1; cat gpurun_out/r02/graph_trace_hist.txt
rm -rf gpurun_out/r02/kt
