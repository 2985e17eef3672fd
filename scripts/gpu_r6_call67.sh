cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_teddy_pair.py -m gpu -q 2>&1 | grep -E "AssertionError|assert|Error|passed|failed" | head -8 | cut -c1-700
