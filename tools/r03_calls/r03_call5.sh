set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03e
mkdir -p $O
timeout 900 python -m pytest tests/test_integer_class.py tests/test_gpu_fuzz_families.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q > $O/pytest_int.txt 2>&1; tail -15 $O/pytest_int.txt
