cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for i in 1 2 3 4 5; do
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapped.py tests/test_zzz_gpu_fold.py -m gpu -q -k "teddy or edge_cases or reference_corpus or wrapped or fold or literal" > gpurun_out/r06_c47_pytest_teddy_$i.log 2>&1; echo pytest=$?; grep -E "AssertionError: |passed|failed" gpurun_out/r06_c47_pytest_teddy_$i.log | cut -c1-600
done
