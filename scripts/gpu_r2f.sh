cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fsm.py -q -x --durations=5 > gpurun_out/pytest_fsm.log 2>&1; echo pytest_fsm=$?; tail -15 gpurun_out/pytest_fsm.log
timeout 600 python scripts/time_patterns.py '\berror\b' '\bGET\b' '\b\d+\.\d+\b' '\b[A-Z]+\b' '\b0x[0-9a-f]+\b' 'error' '\d+\.\d+x?' > gpurun_out/time_look.txt 2>&1; cat gpurun_out/time_look.txt
