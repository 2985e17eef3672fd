cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_parity.py -k "teddy or literal or 8gib or prefix or config or corpus or random" > gpurun_out/r3v_teddy.log 2>&1; echo "teddy tests rc=$?"; tail -6 gpurun_out/r3v_teddy.log
timeout 300 python scripts/time_configs.py 3 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python scripts/time_patterns.py 'HTTP/\d\.\d' 'GET|POST /[a-z]+' 'warning' 'GET|POST|PUT' 2>&1 | grep kernel_ms
timeout 900 python scripts/gpu_fuzz.py 67 500 2>&1 | grep -v amdgpu.ids | tail -3
