cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
timeout 400 python -m pytest tests/test_gpu_wrapped.py -m gpu -q -x > gpurun_out/r06_c49_pytest_wrapped_$i.log 2>&1; echo pytest=$?; grep -E "AssertionError: |passed|failed" gpurun_out/r06_c49_pytest_wrapped_$i.log | cut -c1-700
done
