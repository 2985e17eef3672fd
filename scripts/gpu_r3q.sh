cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_parity.py -k "stops_the_wave or use_both" > gpurun_out/r3q_limit.log 2>&1; echo "limit rc=$?"; tail -12 gpurun_out/r3q_limit.log
timeout 300 python scripts/time_patterns.py '\d+\.\d+\.\d+\.\d+\b' '(GET|POST|PUT)\b /[a-z/]+ HTTP' '\b\w+=\w+;\w+=\w+\b' '\b(\w+)=(\w+)\b' '(?m)^(\d+) (\w+)' '\d+:\d+:\d+' '\d+\.\d+' 2>&1 | grep kernel_ms > gpurun_out/r3q_time_patterns.txt; cat gpurun_out/r3q_time_patterns.txt
