cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_text_anchor.py tests/test_gpu_fsm.py tests/test_zz_gpu_look_wider.py tests/test_gpu_wrapped.py tests/test_zzzz_gpu_routing.py -m gpu -q -x --durations=5 > gpurun_out/r04_text_pytest.log 2>&1; echo pytest=$?; tail -14 gpurun_out/r04_text_pytest.log | cut -c1-500
IP='(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)'
{ CXG_NO_WRAPPED_LITERALS=1 timeout 200 python scripts/time_patterns.py "$IP" '\berror\b' '(^|\s)error' '(?:^|,)\d+' '(^|\s)(GET|POST)' 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c150-
} > gpurun_out/r04_time_text.txt 2>&1; cat gpurun_out/r04_time_text.txt
