set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03j
python tools/graph_launch_overhead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03j/graph_launch_overhead.txt
SMR_FUZZ_SEED_OFFSET=5000 timeout 1500 python -m pytest tests/test_gpu_fuzz_families.py tests/test_gpu_fuzz.py tests/test_integer_class.py -m gpu -q > gpurun_out/r03j/fuzz_seed5000.txt 2>&1; grep -E "passed|failed|Error|FAILED|fuzz families\]" gpurun_out/r03j/fuzz_seed5000.txt | tail -20
