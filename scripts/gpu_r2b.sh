cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fsm.py -q -x > gpurun_out/pytest_fsm.log 2>&1; echo pytest_fsm=$?; tail -25 gpurun_out/pytest_fsm.log
timeout 300 python scripts/time_patterns.py > gpurun_out/time_patterns.log 2>&1; echo tp=$?; tail -30 gpurun_out/time_patterns.log
