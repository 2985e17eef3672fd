cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -22 gpurun_out/pytest_gpu.log
