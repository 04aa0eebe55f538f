#!/bin/bash
mkdir -p gpurun_out/c28
timeout 600 python tools/flat2_ab.py > gpurun_out/c28/flat2_ab.txt 2>&1
cat gpurun_out/c28/flat2_ab.txt | tail -52
