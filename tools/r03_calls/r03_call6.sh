set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03f
mkdir -p $O
for mode in inorder chains; do for k in 20 2000; do
  python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu --no-extra --step-mode $mode > $O/bench_${mode}_$k.json 2> $O/bench_${mode}_$k.err
  python -c "import json,sys; d=json.load(open('$O/bench_${mode}_$k.json')); print('$mode K=$k', d['value'], d['ms_per_step']*1e3, d['config']['step_us_long_graph'], d['roofline']['per_kernel']['permutedims']['us'], d['roofline']['per_kernel']['broadcast4']['us'])"
done; done
timeout 300 python -m pytest tests/test_integer_class.py -m gpu -q 2>&1 | tail -3
