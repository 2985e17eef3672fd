# round 2, call A: full GPU parity tier (with durations), CPU inventory, baseline bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
nproc > gpurun_out/host.txt; lscpu | head -25 >> gpurun_out/host.txt; free -g >> gpurun_out/host.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=25 -x > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -45 gpurun_out/pytest_gpu.log
timeout 300 python bench.py > gpurun_out/bench.log 2>&1; echo bench=$?; tail -1 gpurun_out/bench.log | cut -c1-1500
