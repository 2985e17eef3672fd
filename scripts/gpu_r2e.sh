cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -12 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo bench=$?; cat gpurun_out/bench_cfg3.json
