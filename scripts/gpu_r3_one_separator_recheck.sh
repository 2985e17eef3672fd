cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 80 python -m pytest tests/test_gpu_fields.py tests/test_gpu_trio.py -m gpu -q -k "ab_switches or one_separator or edges or synthlog" 2>&1 | tail -5 | tee gpurun_out/r03_pytest_gpu_recheck.log
