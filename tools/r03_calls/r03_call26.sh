#!/bin/bash
mkdir -p gpurun_out/c26
timeout 600 python tools/reduce_all_sweep.py > gpurun_out/c26/reduce_all_sweep.txt 2>&1
cat gpurun_out/c26/reduce_all_sweep.txt | tail -20
