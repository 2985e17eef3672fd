cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 85 python -m pytest tests/test_gpu_wide.py -m gpu -q -x 2>&1 | tail -25 | cut -c1-400 | tee gpurun_out/r03_pytest_gpu_wide.log
