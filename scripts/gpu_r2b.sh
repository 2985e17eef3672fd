cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fsm.py -q -x > gpurun_out/pytest_fsm.log 2>&1; echo pytest_fsm=$?; tail -25 gpurun_out/pytest_fsm.log
timeout 300 python scripts/time_patterns.py '(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)' '(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]' '\d+\.\d+x?' 'a+b|b+a' '\d+(\.\d+)?%' 'GET|POST /[a-z]+x?' > gpurun_out/time_patterns.log 2>&1; echo tp=$?; tail -30 gpurun_out/time_patterns.log
