cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_trio.py -q -x > gpurun_out/r3y_trio.log 2>&1; echo "trio rc=$?"; tail -12 gpurun_out/r3y_trio.log | cut -c1-300



