# what the driver runs at the end of a round, plus the pattern timings: GPU test tier, smoke, default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -9 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo bench=$?; python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'])"
