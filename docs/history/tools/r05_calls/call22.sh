#!/bin/bash
# round 5, call 22: the memory system's ceiling for the orbit kernel's access pattern (32-KiB boxes with rows of 64 B ... 1 KiB)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 300 tools/bin/run_length_ceiling 128 144 96 2>&1 | tee gpurun_out/run_length_ceiling.txt
