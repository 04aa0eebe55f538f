#!/bin/bash
# The CPU test-suite against the library's host side built with AddressSanitizer + UBSan (planner, canonicalisation and the integer
# typing pass, sequence analysis, metadata parser, runtime-compilation driver, the multi-rank path over gloo).  No GPU needed.
# Usage: bash tools/asan_cpu_suite.sh   (builds /tmp/smr_asan/libstrided_hip_asan.so first; ~5 min)
set -eu
cd "$(dirname "$0")/.."
make -C strided.jl_amd/csrc asan > /tmp/smr_asan_build.log 2>&1 || { tail -20 /tmp/smr_asan_build.log; exit 1; }
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 SMR_LIB=/tmp/smr_asan/libstrided_hip_asan.so \
    python -m pytest tests -q -m "not gpu" --deselect tests/test_kmeta.py -p no:cacheprovider 2>&1 | tail -6
