#!/bin/bash
# r04 call 24: the fuzz-family file alone (its own process, like the driver's single process), then the files behind it in collection order
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fuzz_families.py -x -q -m gpu -rf -p no:cacheprovider 2>&1 | tail -5 | cut -c1-600 | tee $O/pytest_gpu_fuzzfam.txt
timeout 1200 python -m pytest tests -q -m gpu -rf -p no:cacheprovider --deselect tests/test_gpu_fuzz_families.py --deselect tests/test_gpu_fuzz.py --deselect tests/test_gpu_fullsize.py -n 3 2>&1 | tail -6 | cut -c1-400 | tee $O/pytest_gpu_rest.txt
