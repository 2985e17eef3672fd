cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_teddy_pair.py tests/test_gpu_wrapped.py -m gpu -q > gpurun_out/r06_c57_pytest_pair.log 2>&1; echo pytest=$?; grep -E "AssertionError|Error|passed|failed|ROUTING|^\(" gpurun_out/r06_c57_pytest_pair.log | head -20 | cut -c1-600
