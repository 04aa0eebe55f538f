#!/bin/bash
mkdir -p gpurun_out/c36
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c36/smoke.log 2>&1; tail -1 gpurun_out/c36/smoke.log
timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/c36/bench.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/c36/bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
