cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fsm.py -x -q -k "submatch or captures or random_patterns" 2>&1 | tail -3
timeout 300 python scripts/time_patterns.py '(a|ab)(c|bcd)' '(\d+)(\.\d+)?%' '([a-z]+)=(\d+|[a-z])' '(\w+)@(\w+)\.(\w+)' 2>&1 | grep kernel_ms | cut -c1-160
